import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "turbo-range-coder_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    """say LOUDLY which checker binaries are present: without oracle/_ref (a fresh clone on a box that has no /root/reference to build it
    from) the oracle-vs-reference fuzz and the reference-harness / reference-file-format interop tests SKIP, and bench.py's
    cpu_baseline is the oracle port (kind "port"), not the reference (kind "reference")"""
    ref = os.path.join(ROOT, "oracle", "_ref")
    have = [f for f in ("libtrc_ref.so", "turborc_hip", "turborc_ref") if os.path.exists(os.path.join(ref, f))]
    if len(have) == 3:
        return "oracle/_ref: reference build present (libtrc_ref.so, turborc_hip, turborc_ref): reference-backed tests run"
    return ("oracle/_ref: REFERENCE BUILD ABSENT (%s of 3 files) -- tests that compare with the compiled reference SKIP; the oracle "
            "restatement and the committed golden vectors (generated through the reference) still pin parity" % len(have))
