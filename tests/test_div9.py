"""The lean kernel divides its 3 x 3 sums by nine in three operations (q = s r, e = fma(-9, q, s), q' = fma(e, r, q), r = RN(1/9);
mc-cnn_amd/csrc/cbca_lean.hip div9) wherever 2^-95 <= |s| < 2^125, and with the IEEE divide elsewhere.  This walks the float bit
patterns on the CPU (tests/div9_check.c: every 61st pattern + the neighbourhoods of the range limits, both signs; MC_EXHAUSTIVE=1:
all 2^32, ~30 s on 8 cores) and requires the short form to equal s / 9.0f -- the reference's `sum / cnt` (adcensus.cu:375) with
cnt = 9 -- bit for bit on every pattern inside the range."""
import os
import subprocess


def test_three_operation_division_by_nine_is_the_ieee_quotient(tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "div9_check")
    flags = ["-O2", "-fopenmp", "-ffp-contract=off"]
    if " fma " in open("/proc/cpuinfo").read().replace("\n", " "):
        flags.append("-mfma")   # (hardware fma: the same single rounding as libm's fmaf, only faster)
    subprocess.check_call(["gcc"] + flags + [os.path.join(here, "div9_check.c"), "-o", exe, "-lm"])
    stride = "1" if os.environ.get("MC_EXHAUSTIVE") == "1" else "61"
    nin, bad_in, bad_out = (int(v) for v in subprocess.check_output([exe, stride], timeout=900).split())
    assert nin > (1 << 25) and bad_in == 0, (nin, bad_in, bad_out)
    # the kernel's range test is what keeps the few mismatching patterns outside (overflowing intermediates near FLT_MAX, denormal results)
    src = open(os.path.join(here, "..", "mc-cnn_amd", "csrc", "cbca_lean.hip")).read()
    assert "0x1.c71c72p-4f" in src and "a >= 0x1p-95f && a < 0x1p125f" in src
