"""`python bench.py --gpus N` without a launcher: count the GPUs from sysfs and start the N ranks as fresh children.
Nothing in this module may touch the GPU runtime (tests/test_bench_gpu.py checks the source for it)."""
import os
import socket
import subprocess
import sys
import time

BENCH_PY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def visible_gpu_count():
    """GPUs this process's children will see, WITHOUT loading a GPU runtime in this process: the KFD topology in sysfs
    (a readable node with SIMDs is a GPU), else the render nodes in /dev/dri; narrowed by HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.  None when neither can be read.  ADVISORY: the launcher only
    warns on it -- a rank whose device ordinal does not exist fails in `Env` and the launcher propagates its exit code."""
    n = None
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:                                       # 1. KFD topology: a node with SIMDs is a GPU; a container sees the nodes of
        k = 0                                  #    the whole host but can only READ the properties of its own GPUs
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as f:
                    props = dict(l.split()[:2] for l in f if len(l.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                k += 1
        n = k
    except (OSError, ValueError):
        pass
    if not n:                                  # 2. the render nodes the process was given
        try:
            n = len([d for d in os.listdir("/dev/dri") if d.startswith("renderD")])
        except OSError:
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (this process has
    made no GPU call and makes none -- the devices are counted from sysfs, not through the runtime), each with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, let rank 0 print the one JSON line on the inherited stdout, and
    return the first non-zero exit code (the remaining children are then terminated by their own PIDs) or 0."""
    ndev = visible_gpu_count()
    if ndev is not None and ndev < n and "BENCH_FORCE_DEVICE" not in os.environ:
        print(f"bench.py: --gpus {n} but sysfs shows {ndev} GPU(s); starting the ranks anyway (each checks its own device)",
              file=sys.stderr)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this pool
        procs.append(subprocess.Popen([sys.executable, BENCH_PY] + argv, env=env))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.05)
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0:
                rc = code
    for pr in live:                                      # a rank failed: stop the others (exact PIDs we started)
        pr.terminate()
    for pr in live:
        try:
            pr.wait(timeout=20)
        except subprocess.TimeoutExpired:
            pr.kill()
    return rc

