"""CPU: unit test of the range coder's append-only carry scheme (turbo-range-coder_amd/csrc/trc_carry.h -- the
exact header the HIP kernels include) against the reference's write-then-ripple behaviour (turborc_.h:103),
on event streams dense in 0xFFFFFFFF runs and carries through them (tests/carry_host.cpp)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_carry_scheme_matches_ripple():
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "carry_host")
        subprocess.check_call(["g++", "-O1", "-o", exe, os.path.join(ROOT, "tests", "carry_host.cpp")])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.startswith("ok:"), r.stdout
