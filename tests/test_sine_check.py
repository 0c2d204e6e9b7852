"""tools/sine_check.c (CPU only): the exact oscillator's sine decision (modules.hip.h, sine_exact_plain) restated in C against the host libm —
no phase whose rounding the kernels take as decided may differ from `(pos * PI * 2.0).sin() as f32` (oscillator.rs:133)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_decided_sines_are_the_libms(tmp_path):
    exe = str(tmp_path / "sine_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "sine_check.c"), "-lm", "-lpthread"], check=True)
    r = subprocess.run([exe, "3", "4"], capture_output=True, text=True, timeout=300)   # 3 x 12 million phases
    assert r.returncode == 0, r.stdout + r.stderr
    rows = re.findall(r"(\d+) phases: (\d+) decided differently from the libm, (\d+) undecided", r.stdout)
    assert len(rows) == 3 and all(int(n) == 12000000 and int(w) == 0 for n, w, _ in rows), r.stdout
    assert int(rows[0][2]) < 12000000 * 1e-5   # uniform phases: a few in a million take the reference's expression itself


def test_the_restatement_is_the_kernels():
    """the polynomial's coefficients and the decision's constants, as modules.hip.h spells them"""
    src = open(os.path.join(ROOT, "s-rack_amd", "csrc", "modules.hip.h")).read()
    body = src[src.index("SRK_DEV float sine_exact_plain(double pos, bool& cold)"):]
    body = body[:body.index("\n}\n")]
    chk = open(os.path.join(ROOT, "tools", "sine_check.c")).read()
    for token in ("-41.34170223990684", "6.283185307179272", "-76.70584757807868", "81.60524914955879", "-15.081496425342264", "42.05813586028645",
                  "3.6659216216293173", "1.0e-13", "2.0e-15"):
        assert token in body and token in chk, token
