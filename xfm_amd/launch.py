"""One process per GPU: the launch line the reference builds in run.py:44-75 (`python3 -m torch.distributed.launch --nproc_per_node
... --use_env SCRIPT ...`, run through os.system), restated for this stack.

The parent never touches the GPU (no HIP call, no torch.cuda.is_available()): it only starts `python -m torch.distributed.run`
as a CHILD process -- on this pool a process that has initialised HIP must not exec -- waits for it and hands its exit code
back.  Rendezvous is always 127.0.0.1 (container hostnames may not resolve)."""
import os
import socket
import subprocess
import sys


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _visible_filter(n):
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [t for t in v.split(",") if t.strip() != ""]
            return min(n, len(ids))
    return n


def visible_gpu_count():
    """Number of GPUs WITHOUT loading the HIP runtime: the KFD topology lists one node per agent, and GPU nodes are the ones with
    SIMDs (`simd_count` > 0 in .../nodes/N/properties); *_VISIBLE_DEVICES narrows it.  Falls back to torch's device_count() (which
    does not create a context on this ROCm build of torch) only where the topology is not readable."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        n += int(line.split()[1]) > 0
                        break
        if n > 0:
            return _visible_filter(n)
    except OSError:
        pass
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def launch_command(script, script_args, nproc, master_port=None, nnodes=1, node_rank=0, master_addr="127.0.0.1", module=False):
    port = master_port or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", f"--nnodes={nnodes}", f"--nproc-per-node={nproc}",
           "--master-addr", master_addr, "--master-port", str(port)]
    if nnodes > 1:
        cmd += [f"--node-rank={node_rank}"]
    cmd += (["-m", script] if module else [script]) + [str(a) for a in script_args]
    return cmd


def launch(script, script_args, nproc, master_port=None, env=None, visible_devices=None, **kw):
    """Start `nproc` ranks of `script` on this node and wait.  Returns the launcher's exit code (non-zero when any rank failed)."""
    cmd = launch_command(script, script_args, nproc, master_port, **kw)
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    e.setdefault("OMP_NUM_THREADS", "4")
    e["MASTER_ADDR"] = kw.get("master_addr", "127.0.0.1")  # torchrun re-exports it to the ranks from --master-addr
    if visible_devices is not None:
        e["HIP_VISIBLE_DEVICES"] = visible_devices
        e["CUDA_VISIBLE_DEVICES"] = visible_devices
    if env:
        e.update(env)
    print("### launch:", " ".join(cmd), file=sys.stderr, flush=True)   # stderr: a caller's stdout may be a one-line JSON contract (bench.py)
    return subprocess.call(cmd, env=e)
