"""K3 (csrc/vga_map.hip, vga_chain_dp) turns the winning candidate's integer score round(1000 s) back into f(i) = x / 1000 with two
fused multiply-adds instead of the f64 division sequence.  That is only allowed because the result is the correctly rounded quotient
for EVERY 32-bit integer x: checked here exhaustively, in C (the same three operations, contraction off), on the host."""
import os
import subprocess

SRC = r"""
#include <math.h>
#include <stdio.h>
int main(void)
{
    const double c = 1.0 / 1000.0;
    long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long x = -2147483648L; x <= 2147483647L; x++) {
        const double xd = (double)x;
        const double want = xd / 1000.0 + 0.0;
        const double q = xd * c;
        const double r = fma(-q, 1000.0, xd);
        const double got = fma(r, c, q) + 0.0;
        if (got != want || signbit(got) != signbit(want)) bad++;
    }
    printf("%ld\n", bad);
    return 0;
}
"""


def test_division_by_1000_through_fma_is_exact_for_every_int32(tmp_path):
    src = os.path.join(str(tmp_path), "div1000.c")
    exe = os.path.join(str(tmp_path), "div1000")
    open(src, "w").write(SRC)
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-mfma", src, "-o", exe, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="8"))
    assert out.returncode == 0 and out.stdout.strip() == "0", out.stdout + out.stderr
