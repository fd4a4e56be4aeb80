"""Host-memory guard for every python process that runs on a GPU box (bench.py, pytest).

Round 3 lost three GPU boxes to a 10^6 x 10^6 numpy broadcast in a test helper (DESIGN.md "Incident"): the boxes
over-commit, so the allocation succeeded and the box died while the pages were touched.  This module bounds the
process instead of trusting every array expression:

  * a watchdog thread samples /proc/self/statm every 20 ms and ends the process (exit 137, message on stderr) once
    the resident set passes the cap.  numpy releases the GIL inside its element loops, which is where such a bomb
    spends its time, so the thread gets to run; a box fills at ~10 GB/s, i.e. ~0.2 GB of overshoot.
  * optionally (MW_MEMGUARD=data | as) a kernel-enforced RLIMIT_DATA / RLIMIT_AS.  RLIMIT_AS cannot be the default:
    the ROCm runtime reserves terabytes of address space (SVM apertures) at initialisation, which an address-space
    limit of a few dozen GB refuses (measured in round 4: profiles/r04_memguard_probe.txt).

MW_HOST_MEM_CAP_GB (default 48) sets the cap, MW_MEMGUARD=off disables the guard.

The cap is per PROCESS.  Under a launcher with several local ranks (torch.distributed.run sets LOCAL_WORLD_SIZE) the default cap is
divided by the number of local ranks (not below 12 GiB), so that the ranks TOGETHER stay near the allowance of one process; an explicit
MW_HOST_MEM_CAP_GB is taken as given.  When the watchdog ends a rank it does so with os._exit: no HIP / RCCL teardown runs, and peer
ranks blocked in a collective stay blocked until their own timeout (bench.py bounds the communicator bootstrap with
MW_BENCH_TILES_TIMEOUT; the driver bounds the run) -- a dead box would have been worse.
"""
import os
import sys
import threading

_installed = False


def _rss_bytes() -> int:
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def install(cap_gb: float | None = None) -> str:
    """Idempotent.  Returns the mode that is active ("watchdog", "data", "as", "off")."""
    global _installed
    mode = os.environ.get("MW_MEMGUARD", "watchdog")
    if _installed or mode == "off":
        return mode
    _installed = True
    if cap_gb is None and "MW_HOST_MEM_CAP_GB" not in os.environ:
        try:
            local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        except ValueError:
            local = 1
        cap_gb = max(12.0, 48.0 / local)      # (a rank with torch + the ROCm runtime + the parity gate's f64 arrays sits at 3-5 GiB)
    cap = int(float(cap_gb if cap_gb is not None else os.environ["MW_HOST_MEM_CAP_GB"]) * 2**30)
    if mode in ("data", "as"):
        import resource
        which = resource.RLIMIT_DATA if mode == "data" else resource.RLIMIT_AS
        _, hard = resource.getrlimit(which)
        resource.setrlimit(which, (cap if hard == resource.RLIM_INFINITY else min(cap, hard), hard))

    def watch():
        import time
        while True:
            try:
                rss = _rss_bytes()
            except Exception:
                return
            if rss > cap:
                os.write(2, (f"\nmemguard: resident set {rss / 2**30:.1f} GiB passed the cap of {cap / 2**30:.0f} GiB "
                             f"(MW_HOST_MEM_CAP_GB) -- ending pid {os.getpid()} before the box does\n").encode())
                os._exit(137)
            time.sleep(0.02)

    threading.Thread(target=watch, name="memguard", daemon=True).start()
    return mode


if __name__ == "__main__":  # self-test: python tests/memguard.py  -> must exit 137 quickly
    os.environ.setdefault("MW_HOST_MEM_CAP_GB", "1")
    install()
    import numpy as np
    a = np.arange(40_000, dtype=np.float32)
    print(np.abs(a[:, None] - a[None, :]).sum())  # 6.4 GB of temporaries
    sys.exit(0)
