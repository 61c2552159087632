"""bench.py's contract is ONE JSON line on stdout; libraries (RCCL's version banner when a communicator is created) and child
processes write to descriptor 1 too.  claim_stdout() points descriptor 1 at stderr and keeps a private duplicate for the line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = """
import os, sys, ctypes
sys.path.insert(0, %r)
import bench
bench.claim_stdout()
ctypes.CDLL(None).puts(b"banner from a C library")      # C stdio on descriptor 1, buffered until exit like RCCL's
os.write(1, b"raw write to descriptor 1\\n")
print("python print")
bench.emit({"metric": "m", "value": 1.5})
"""


def test_stdout_carries_exactly_the_json_line():
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "m", "value": 1.5}, r.stdout
    assert "banner from a C library" in r.stderr and "raw write" in r.stderr and "python print" in r.stderr
