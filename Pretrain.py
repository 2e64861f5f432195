"""Pre-training task script on the HIP hot path: the `main()` of the reference's Pretrain.py:306-476 (argparse + YAML config ->
model, optimizer, scheduler, accelerator, resume, train loop, checkpoints), started one process per GPU by run.py.

What differs, on purpose: the dataset package (JSON-lines / HDFS readers, tokenisers, PIL augmentation) is outside the hot-path
scope, so the loaders here are the synthetic generators of SURVEY 8(d) (`config['synthetic']`, the default when `train_file` is
empty as in the shipped YAML); a caller with real loaders passes them to `xfm_amd.pretrain_loop.train` directly.
"""
import argparse
import json
import math
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import yaml  # noqa: E402


class SyntheticLoader:
    """`steps` batches in the tuple layout of dataset/pretrain_dataset.py:264-312's collate: (image, text_ids, text_atts,
    text_ids_masked, masked_pos, masked_ids) -- or without the image for the text source."""

    def __init__(self, steps, batch_size, seed, image_res=224, max_tokens=30, max_masks=15, with_image=True, vocab=None, pool=4):
        from xfm_amd import synthetic as syn
        kw = {} if vocab is None else {"vocab": vocab}
        self.steps = steps
        self.batches = []
        for k in range(min(pool, steps)):  # a small pool of distinct batches, cycled (host generation is not what is being run)
            b = syn.pretrain_batch(batch_size, seed=seed + 7919 * k, image_res=image_res, max_tokens=max_tokens, max_masks=max_masks,
                                   with_image=with_image, **kw)
            t = (b["text_ids"], b["text_atts"], b["text_ids_masked"], b["masked_pos"], b["masked_ids"])
            self.batches.append(((b["image"],) + t) if with_image else t)

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield self.batches[i % len(self.batches)]


class Checkpointer:
    """utils/checkpointer.py:20-47, local paths only."""

    def __init__(self, serialization_dir=".output"):
        self._dir = serialization_dir
        os.makedirs(self._dir, exist_ok=True)

    def save_checkpoint(self, epoch, model_state, training_states, step=-1):
        if step > 0:
            torch.save(model_state, os.path.join(self._dir, "model_state_step_{}.th".format(step)))
        else:
            torch.save(model_state, os.path.join(self._dir, "model_state_epoch_{}.th".format(epoch)))
            torch.save({**training_states, "epoch": epoch}, os.path.join(self._dir, "training_state_latest.th"))


def main(args, config):
    from xfm_amd import pretrain_loop as PL
    from xfm_amd.accelerators import ACCELERATOR_MAP
    from xfm_amd.model_pretrain import XFM

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world_size = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise RuntimeError("Pretrain.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world_size > 1 and not dist.is_initialized():  # utils.init_distributed_mode (utils/__init__.py:388-410)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", world_size=world_size, rank=rank)

    config["batch_size"] = config["images"]["batch_size"]
    if args.bs > 0:
        config["batch_size"] = config["images"]["batch_size"] = args.bs
    if args.epoch > 0:
        config["schedular"]["epochs"] = args.epoch
        print(f"### set epochs to: {args.epoch}", flush=True)
    seed = args.seed + rank  # Pretrain.py:333
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)

    if config.get("train_file") and not config.get("synthetic", False):
        raise NotImplementedError("file-backed datasets (dataset/pretrain_dataset.py) are outside the hot-path scope: set "
                                  "`synthetic: true` or drive xfm_amd.pretrain_loop.train with your own loaders")
    step_per_epoch = math.ceil(config["train_dataset_size"] / (config["batch_size"] * world_size))
    steps = step_per_epoch * config["schedular"]["epochs"]
    vocab = config.get("text_config", {}).get("vocab_size")
    mk = dict(image_res=config["image_res"], max_tokens=config.get("max_tokens", 30), max_masks=config.get("max_masks", 15), vocab=vocab)
    image_loader = SyntheticLoader(steps, config["batch_size"], seed, **mk)
    text_loader = None
    if config.get("synthetic_text_source", False):  # stands in for train_file_text (run_text_iter: a text-only MLM step)
        t = config.get("texts", {})
        text_loader = SyntheticLoader(steps, t.get("batch_size", config["batch_size"]), seed + 1, with_image=False,
                                      max_tokens=min(t.get("max_tokens", 30), 128), max_masks=t.get("max_masks", 15), vocab=vocab,
                                      image_res=config["image_res"])

    print("Creating model XFM", flush=True)
    model = XFM(config=config).to(device)
    arg_opt = PL.AttrDict(config["optimizer"])
    optimizer = PL.create_optimizer(arg_opt, model)
    arg_sche = PL.AttrDict(config["schedular"])
    arg_sche["step_per_epoch"] = step_per_epoch
    lr_scheduler = PL.create_scheduler(arg_sche, optimizer)
    arg_acc = PL.AttrDict(config["accelerator"])
    accelerator = ACCELERATOR_MAP[arg_acc["ACCELERATOR"]](arg_acc, logger=None)

    start_epoch = 0
    if config.get("resume", False):  # Pretrain.py:437-441
        start_epoch = PL.resume(args.checkpoint, optimizer, lr_scheduler)
    if args.checkpoint and os.path.exists(args.checkpoint):
        model.load_pretrained(args.checkpoint, config, is_domain_pretrain=True)
    model, optimizer, lr_scheduler = accelerator.set_up(model, optimizer, lr_scheduler, local_rank, world_size, rank)
    checkpointer = Checkpointer(args.output_dir)
    print("### output_dir, ", args.output_dir, flush=True)

    def log(step, avg):
        if rank == 0:
            print(json.dumps({"step": step, **{k: round(v, 5) for k, v in avg.items()}}), flush=True)

    start_time = time.time()
    print("Start training", flush=True)
    stats = PL.train(model, image_loader, (None, None, None, None, text_loader), optimizer, (start_epoch, config["schedular"]["epochs"]),
                     device, lr_scheduler, config, accelerator, checkpointer, world_size=world_size,
                     print_freq=config.get("print_freq", 50), log=log)
    torch.cuda.synchronize()
    if world_size > 1:
        dist.barrier()
    if rank == 0:
        with open(os.path.join(args.output_dir, "log.txt"), "a") as f:
            f.write(json.dumps({**{f"train_{k}": v for k, v in stats.items()}, "epochs": config["schedular"]["epochs"]}) + "\n")
        with open(os.path.join(args.output_dir, "config.yaml"), "w") as f:
            yaml.safe_dump(config, f)
        print("### Time {:.1f} s".format(time.time() - start_time), flush=True)
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--checkpoint", type=str, default="")
    parser.add_argument("--config", type=str, required=True)
    parser.add_argument("--output_dir", type=str, default="output/pretrain")
    parser.add_argument("--device", default="cuda")
    parser.add_argument("--seed", default=42, type=int)
    parser.add_argument("--epoch", default=-1, type=int)
    parser.add_argument("--bs", default=-1, type=int)
    parser.add_argument("--distributed", action="store_false")
    a = parser.parse_args()
    with open(a.config) as f:
        cfg = yaml.safe_load(f)
    os.makedirs(a.output_dir, exist_ok=True)
    main(a, cfg)
