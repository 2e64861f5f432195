"""Launcher with the reference's command line (run.py:289-392: --task --dist --config --output_dir --checkpoint --bs --seed --epoch
--master_port) -> one process per GPU on this node.

    python3 run.py --task pretrain_DIY --dist 1 --config configs/Pretrain_synthetic.yaml --output_dir output/pt

`--dist` follows run.py:44-75: '1' / 'all' = every visible GPU of this node, 'f4' / 'l4' = the first / last four, 'gpuK' = GPU K
alone.  The reference builds a `torch.distributed.launch --use_env` shell line and hands it to os.system; here the ranks are started
by xfm_amd.launch (torch.distributed.run as a child process, rendezvous on 127.0.0.1) and the launcher's exit code is returned.
Tasks whose scripts are outside the hot-path scope (captioning, grounding, NLVR data handling ...) raise NotImplementedError, as the
reference does for unknown tasks (run.py:352)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from xfm_amd.launch import launch, launch_command, visible_gpu_count  # noqa: E402

TASK_SCRIPTS = {"pretrain_DIY": "Pretrain.py"}
DEFAULT_CONFIGS = {"pretrain_DIY": "configs/Pretrain_synthetic.yaml"}


def get_dist(args, n_visible=None):
    """(nproc, visible-device string or None) for --dist (run.py:44-75)."""
    n = visible_gpu_count() if n_visible is None else n_visible
    d = args.dist
    if d in ("1", "all"):
        return max(n, 1), None
    if d == "f4":
        return 4, "0,1,2,3"
    if d == "l4":
        return 4, "4,5,6,7"
    if d.startswith("gpu"):
        num = int(d[3:])
        assert 0 <= num <= 8
        return 1, str(num)
    raise ValueError(f"--dist {d}")


def task_command(args, n_visible=None):
    """-> (argv of the launch, visible-device string) without starting anything (also what the tests look at)."""
    if args.task not in TASK_SCRIPTS:
        raise NotImplementedError(f"task == {args.task}")
    if not args.config or not os.path.exists(args.config):
        args.config = os.path.join(ROOT, DEFAULT_CONFIGS[args.task])
    nproc, vis = get_dist(args, n_visible)
    script_args = ["--seed", args.seed, "--epoch", args.epoch, "--config", args.config, "--output_dir", args.output_dir]
    if args.bs > 0:
        script_args += ["--bs", max(args.bs // nproc, 1)]  # "for each gpu, batch_size = bs // num_gpus" (run.py:362-363)
    if args.checkpoint:
        script_args += ["--checkpoint", args.checkpoint]
    return launch_command(os.path.join(ROOT, TASK_SCRIPTS[args.task]), script_args, nproc, args.master_port), nproc, vis, script_args


def run(args):
    _, nproc, vis, script_args = task_command(args)
    print(f"### Start {args.task} on {nproc} GPU(s)", flush=True)
    return launch(os.path.join(ROOT, TASK_SCRIPTS[args.task]), script_args, nproc, master_port=args.master_port, visible_devices=vis)


def parse(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--task", type=str, required=True)
    parser.add_argument("--dist", type=str, required=True, help="see get_dist")
    parser.add_argument("--config", default="", type=str, help="if not given, use default")
    parser.add_argument("--model", default="xfm-ft", type=str)
    parser.add_argument("--epoch", default=-1, type=int)
    parser.add_argument("--bs", default=-1, type=int)
    parser.add_argument("--checkpoint", default="", type=str)
    parser.add_argument("--load_ckpt_from", default="", type=str)
    parser.add_argument("--output_dir", type=str, required=True)
    parser.add_argument("--output_hdfs", type=str, default="")
    parser.add_argument("--evaluate", action="store_true")
    parser.add_argument("--seed", default=42, type=int)
    parser.add_argument("--master_port", default=12345, type=int)
    return parser.parse_args(argv)


if __name__ == "__main__":
    a = parse()
    os.makedirs(a.output_dir, exist_ok=True)
    sys.exit(run(a))
