"""GAF-vs-GAF node-path agreement, the metric of the reference's evaluation scripts
(experiments-snakemake/gafcompare.py:27-78, restated without pandas):

  for every record of the REFERENCE GAF whose read name also occurs in the evaluated GAF (its first record with
  that name): node paths are parsed from column 6 with the pattern (>|<)([0-9]+) into signed ids (+id for '>',
  -id for '<'); identical lists score 1.0; otherwise the score is the Jaccard index of the two integer ranges
  [min, max) of the ids:  |range(max(mins), min(maxes))| / |range(min(mins), max(maxes))|  (0 when the union is empty).
  Reported: matching reads, the average, and the per-read list.

One deviation, stated: a record without any node (the placeholder path '*') makes the reference script raise on
min() of an empty list; here it scores 0.0.

    python -m rs-vgaligner_amd.gafcompare mine.gaf truth.gaf         (run as a module through __graft_entry__.load_package)
"""
from __future__ import annotations

import re
import sys
from typing import Dict, List, Tuple

_NODE = re.compile(r"(>|<)([0-9]+)")


def signed_path(path_field: str) -> List[int]:
    return [int(n) if o == ">" else -int(n) for o, n in _NODE.findall(path_field)]


def jaccard(mine: List[int], ref: List[int]) -> float:
    if mine == ref and mine:
        return 1.0
    if not mine or not ref:
        return 0.0
    inter = range(max(min(mine), min(ref)), min(max(mine), max(ref)))
    union = range(min(min(mine), min(ref)), max(max(mine), max(ref)))
    return len(inter) / len(union) if len(union) else 0.0


def read_gaf_paths(text: str) -> List[Tuple[str, str]]:
    """(read name, path field) per record, in file order"""
    out = []
    for ln in text.splitlines():
        if ln:
            f = ln.split("\t")
            out.append((f[0], f[5] if len(f) > 5 else "*"))
    return out


def compare(mine_text: str, ref_text: str) -> Dict:
    mine: Dict[str, str] = {}
    for name, path in read_gaf_paths(mine_text):
        mine.setdefault(name, path)  # the first record with that name
    scores, found = [], 0
    ref = read_gaf_paths(ref_text)
    for name, path in ref:
        if name in mine:
            found += 1
            scores.append(jaccard(signed_path(mine[name]), signed_path(path)))
    return {"matching_reads": found, "total_ref_reads": len(ref), "avg_jaccard": sum(scores) / len(scores) if scores else 0.0,
            "jaccard": scores}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        raise SystemExit("usage: gafcompare GAF1 REF")
    r = compare(open(argv[0]).read(), open(argv[1]).read())
    print("Matching reads: {}/{}".format(r["matching_reads"], r["total_ref_reads"]))
    print("AVG Jaccard is: {}".format(r["avg_jaccard"]))
    print("Jaccard list is: \n {}".format(",".join(str(v) for v in r["jaccard"])))


if __name__ == "__main__":
    main()
