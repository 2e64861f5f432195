"""Host logic of the pre-training task loop (xfm_amd.pretrain_loop): optimizer groups and the linear schedule against fixtures taken
from the reference's optim.py / scheduler.py (tools/oracle/gen_golden.py::gen_harness), and the multi-source step sequence of
Pretrain.py:211-243 against a hand-restated expectation.  Runs without a GPU."""
import numpy as np
import torch

from golden_util import load
from xfm_amd import pretrain_loop as PL


def _cfg(meta):
    return {"use_beit_v2": True, "image_res": 224, "patch_size": 16, "local_attn_depth": -1, "text_encoder": "roberta-base",
            "text_num_hidden_layers": meta["text_layers"], "text_fusion_start_at": meta["text_layers"],
            "fusion_num_hidden_layers": meta["fusion_layers"], "fusion_fusion_start_at": 0, "embed_dim": 256, "temp": 0.07,
            "learnable_temp": True, "max_temp": 0.5, "min_temp": 0.001}


def test_optimizer_groups_match_reference():
    from xfm_amd.model_pretrain import XFM
    z, meta = load("harness")
    with torch.device("meta"):
        m = XFM(_cfg(meta))
    assert sorted(m.init_params) == sorted(meta["init_params"])
    groups = PL.optimizer_groups(m)
    for gi in range(4):
        assert groups[gi] == meta["groups"][gi], (gi, set(groups[gi]) ^ set(meta["groups"][gi]))
    # substring rules, as in the reference: the layer-scale gammas and the temperature DO decay (no rule names them), while
    # `relative_position_bias_table` does not (it contains "bias")
    decayed = groups[0] + groups[2]
    assert any("gamma_1" in n for n in decayed) and "temp" in decayed
    assert any("relative_position_bias_table" in n for n in groups[1])


def test_create_optimizer_and_linear_schedule_match_reference():
    z, meta = load("harness")
    w = [torch.nn.Parameter(torch.zeros(3)) for _ in range(4)]

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(2, 2)          # a.weight -> decay, a.bias -> no decay
            self.head = torch.nn.Linear(2, 2)       # init_params -> the two lr_mult groups
            self.init_params = ["head.weight", "head.bias"]

    m = Tiny()
    opt = PL.create_optimizer(PL.AttrDict(lr=1e-4, weight_decay=0.01, lr_mult=2), m)
    got = [[g["lr"], g["weight_decay"], list(g["betas"]), g["eps"]] for g in opt.param_groups]
    assert got == meta["hyper"]
    assert [len(g["params"]) for g in opt.param_groups] == [1, 1, 1, 1]
    sch = PL.create_scheduler(PL.AttrDict(sched="linear", num_warmup_steps=0.1, epochs=2, step_per_epoch=10), opt)
    lrs = []
    for _ in range(23):
        lrs.append([opt.param_groups[0]["lr"], opt.param_groups[2]["lr"]])
        opt.step()
        sch.step()
    assert np.allclose(np.asarray(lrs), z["lrs"], rtol=1e-12, atol=0.0)
    assert lrs[0][0] == 0.0 and lrs[2][0] == 1e-4 and lrs[22][0] == 0.0  # warm-up 2 steps, peak, clamped at zero past the end


class _Model(torch.nn.Module):
    def __init__(self, log):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(()))
        self.log = log

    def forward(self, image, text_ids, text_atts, text_ids_masked=None, masked_pos=None, masked_ids=None, ret_match_loss=True,
                ret_mim_loss=True, ret_mlm_loss=True, ret_itc_loss=True, data_source=None):
        self.log.append(("fwd", data_source if image is not None else "text", ret_itc_loss, ret_match_loss, ret_mlm_loss, ret_mim_loss))
        one = self.p * 0 + 1.0
        return {"loss_itc": one * 1, "loss_itm": one * 2, "loss_mlm": one * 3, "loss_mim": one * 4}


class _Acc:
    def __init__(self, log):
        self.log = log

    def backward_step(self, loss, optimizer):
        self.log.append(("bwd", float(loss)))

    def optimizer_step(self, optimizer, model):
        self.log.append(("opt",))


def _batches(n, with_image=True):
    t = torch.zeros(2, 4, dtype=torch.long)
    for _ in range(n):
        yield ([torch.zeros(2, 3, 8, 8)] if with_image else []) + [t, t, t, t, t]


def test_train_loop_source_order_gates_and_optimizer_steps():
    """Pretrain.py:211-243 with text + web + imagenet + image sources, stop_calc_itm = 3, stop_calc_mm = 4:
    per step: text (own optimizer step) -> web (never steps) -> imagenet (steps only once the multimodal losses stopped or without
    a web source) -> image (steps; skipped from global_step 4 on).  global_step starts at 1."""
    log = []
    m, acc = _Model(log), _Acc(log)
    opt = torch.optim.SGD([{"params": [m.p], "lr": 0.5}, {"params": [], "lr": 0.5}, {"params": [], "lr": 1.0}, {"params": [], "lr": 1.0}])
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    cfg = {"train_dataset_size": 100, "batch_size": 2, "stop_calc_itm": 3, "stop_calc_mm": 4, "ckpt_frequent": 1, "ckpt_frequent_step": 10 ** 9}
    saved = []

    class Ckpt:
        def save_checkpoint(self, **kw):
            saved.append(kw)

    out = PL.train(m, _batches(5), (None, _batches(5), _batches(5), None, _batches(5, with_image=False)), opt, (0, 1), "cpu", sch, cfg, acc,
                   checkpointer=Ckpt(), print_freq=2)
    want = []
    for gs in range(1, 6):
        itm, mm = gs < 3, gs < 4
        want += [("fwd", "text", True, True, True, True), ("bwd", 3.0), ("opt",)]
        want += [("fwd", "web", True, itm, True, True), ("bwd", 10.0)]
        want += [("fwd", "imagenet", True, itm, True, True), ("bwd", 10.0)] + ([] if mm else [("opt",)])
        if mm:
            want += [("fwd", "image", True, itm, True, True), ("bwd", 10.0), ("opt",)]
    assert log == want
    assert out["loss_tmlm"] == "3.00000" and out["loss_witc"] == "1.00000" and out["loss_imim"] == "4.00000" and out["loss_mlm"] == "3.00000"
    assert "loss_wmim" in out and "loss_amlm" not in out and out["lr"] == "0.50000" and out["lr_large"] == "1.00000"
    assert saved == []  # step_per_epoch = 50: no epoch boundary within 5 steps


def test_train_loop_aux_source_and_checkpoint_schedule():
    log = []
    m, acc = _Model(log), _Acc(log)
    opt = torch.optim.SGD([{"params": [m.p], "lr": 0.5}, {"params": [], "lr": 0.5}, {"params": [], "lr": 1.0}, {"params": [], "lr": 1.0}])
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    cfg = {"train_dataset_size": 6, "batch_size": 2, "ckpt_frequent": 1, "ckpt_frequent_step": 4}  # 3 steps per epoch
    saved = []

    class Ckpt:
        def save_checkpoint(self, **kw):
            saved.append((kw.get("epoch"), kw.get("step"), sorted(kw["model_state"].keys())))

    PL.train(m, _batches(6), (_batches(6), None, None, None, None), opt, (0, 2), "cpu", sch, cfg, acc, checkpointer=Ckpt())
    # aux batches: MLM (+MIM) only, never an optimizer step of their own (Pretrain.py:229-231)
    assert log[:5] == [("fwd", "aux", False, False, True, True), ("bwd", 10.0), ("fwd", "image", True, True, True, True), ("bwd", 10.0), ("opt",)]
    # (global_step + 1) % 3 == 0 at global_step 2 and 5 -> epoch checkpoints with optimizer / scheduler state; (gs + 1) % 4 == 0 at 3
    assert saved == [(0, None, ["config", "epoch", "lr_scheduler", "model", "optimizer"]), (1, 3, ["config", "model"]),
                     (1, None, ["config", "epoch", "lr_scheduler", "model", "optimizer"])]


def test_train_loop_resumed_run_stops_at_max_epoch():
    """utils.MetricLogger.log_every (driven from Pretrain.py:205-209) ends a run after (max_epoch - start_epoch) * step_per_epoch
    batches: resumed at epoch 1 of 2 with 3 steps per epoch, exactly 3 more steps run even when the loader holds more."""
    log = []
    m, acc = _Model(log), _Acc(log)
    opt = torch.optim.SGD([{"params": [m.p], "lr": 0.5}, {"params": [], "lr": 0.5}, {"params": [], "lr": 1.0}, {"params": [], "lr": 1.0}])
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    cfg = {"train_dataset_size": 6, "batch_size": 2, "ckpt_frequent": 10 ** 9, "ckpt_frequent_step": 10 ** 9}
    PL.train(m, _batches(6), (None, None, None, None, None), opt, (1, 2), "cpu", sch, cfg, acc)
    assert sum(1 for e in log if e == ("opt",)) == 3
