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


def visible_gpu_count():
    """Number of GPUs without initialising the runtime (device_count() does not create a context on ROCm builds of torch)."""
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
    e["MASTER_ADDR"] = "127.0.0.1" if "--master-addr" not in kw else e.get("MASTER_ADDR", "127.0.0.1")
    if visible_devices is not None:
        e["HIP_VISIBLE_DEVICES"] = visible_devices
        e["CUDA_VISIBLE_DEVICES"] = visible_devices
    if env:
        e.update(env)
    print("### launch:", " ".join(cmd), flush=True)
    return subprocess.call(cmd, env=e)
