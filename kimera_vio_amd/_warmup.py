"""First touch of the device in a throw-away process.

On the MI355X pool a process that is the FIRST to create a context on a freshly provisioned box has, a few times per
hundred boxes, died inside HIP initialisation (memory access fault / abort before any kernel of this library ran; the next
process on the same box works).  The test session, `smoke()` and `bench.py` therefore let a child process create and
destroy a small front-end context first and retry a couple of times; nothing is computed there and nothing is caught in
the calling process -- a library that cannot create a context still fails loudly right afterwards."""
from __future__ import annotations

import os
import subprocess
import sys

_CODE = r"""
import os, sys
sys.path.insert(0, %r)
from kimera_vio_amd import frontend as F, params as P
G = os.path.join(%r, "tests", "golden")
L = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
R = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
c = F.Context(L, R, p, batch=1, device=%d)
c.close()
print("warm")
"""


def warm_up_device(attempts: int = 3, timeout_s: int = 180, device: int = 0) -> bool:
    """True when a child process created and destroyed a context on `device`.  A failed attempt is reported on stderr
    (return code and the tail of the child's stderr) -- it is retried, never hidden; the caller's own context creation
    right afterwards is what fails loudly if the device really is unusable."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _CODE % (root, root, int(device))
    for k in range(max(1, attempts)):
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s)
            if r.returncode == 0 and "warm" in r.stdout:
                return True
            print(f"[kvfe warm-up] device {device}, attempt {k + 1}: rc={r.returncode} stderr tail: "
                  f"{(r.stderr or '')[-400:]!r}", file=sys.stderr)
        except subprocess.TimeoutExpired:
            print(f"[kvfe warm-up] device {device}, attempt {k + 1}: no answer within {timeout_s} s", file=sys.stderr)
        except OSError as e:
            print(f"[kvfe warm-up] device {device}, attempt {k + 1}: {e}", file=sys.stderr)
    return False
