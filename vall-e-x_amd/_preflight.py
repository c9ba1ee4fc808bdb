"""Box pre-flight: runs csrc/vx_preflight.bin (library-free HIP program: 8 MiB host -> device -> kernel -> host round trip) in a
SUBPROCESS and collects the facts that identify the lease.  A GPU memory fault aborts the process that caused it; run here, it
becomes a message ("box-level: plain HIP copy faults before any vallex kernel") instead of a dead pytest / smoke process.  What the
check stands in front of: the weight upload that replaces load_state_dict + .to(device), utils/generation.py:79-83."""
from __future__ import annotations

import json
import os
import signal
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "csrc", "vx_preflight.bin")


def run(mode: str = "pinned", timeout: float = 180.0, retries: int = 1) -> dict:
    """{"ok": bool, "mode", "rc", "signal", "detail": <JSON of the program or the tail of its output>, "attempts"}; never raises for a
    dead child.  A child killed by a signal (a GPU memory fault aborts it) is tried `retries` more times after a pause: a fault that
    only hits the first process to touch a fresh lease is absorbed here, and `attempts` > 1 says so in the log."""
    import time
    res = _run_once(mode, timeout)
    n = 1
    while not res["ok"] and res["signal"] not in (None, "timeout") and n <= retries:
        time.sleep(2.0)
        res = _run_once(mode, timeout)
        n += 1
    res["attempts"] = n
    return res


def _run_once(mode: str, timeout: float) -> dict:
    if not os.path.exists(BIN):
        return {"ok": False, "mode": mode, "rc": None, "signal": None,
                "detail": f"{BIN} not built (python -c 'import __graft_entry__ as g; g.build()')"}
    try:
        r = subprocess.run([BIN, mode], capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"ok": False, "mode": mode, "rc": None, "signal": "timeout", "detail": f"no answer within {timeout:.0f} s"}
    sig = signal.Signals(-r.returncode).name if r.returncode < 0 else None
    detail = None
    for line in reversed(r.stdout.strip().splitlines()):
        try:
            detail = json.loads(line)
            break
        except ValueError:
            continue
    if detail is None:
        detail = (r.stdout + r.stderr)[-600:]
    return {"ok": r.returncode == 0 and isinstance(detail, dict) and bool(detail.get("ok")), "mode": mode, "rc": r.returncode,
            "signal": sig, "detail": detail}


def box_facts() -> dict:
    """Strings that identify the lease: ROCm install, kernel driver, host kernel, memlock limit, the runtime knobs that matter."""
    def read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None
    facts = {"rocm": read("/opt/rocm/.info/version"), "amdgpu": read("/sys/module/amdgpu/version"), "kernel": os.uname().release,
             "cpus": os.cpu_count(), "HSA_XNACK": os.environ.get("HSA_XNACK"),
             "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
             "kfd": os.path.exists("/dev/kfd")}
    try:
        import resource
        facts["memlock"] = resource.getrlimit(resource.RLIMIT_MEMLOCK)[0]
    except Exception:
        pass
    return facts


def describe(res: dict) -> str:
    return (f"box-level: plain HIP {res['mode']} copy + kernel failed before any vallex kernel ran "
            f"(rc {res['rc']}, signal {res['signal']}): {res['detail']} | box {box_facts()}")
