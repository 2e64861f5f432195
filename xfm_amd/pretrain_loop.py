"""The pre-training task loop around the HIP step: optimizer / scheduler construction and the multi-source step logic of Pretrain.py.

Mirrors (names, argument meaning, ordering):
  * create_optimizer    optim.py:4-50      4 AdamW groups (decay / no-decay) x (lr / lr * lr_mult for model.init_params), betas (0.9, 0.98)
  * create_scheduler    scheduler.py:4-30  linear warm-up then linear decay (LambdaLR)
  * run_image_iter / run_text_iter / train   Pretrain.py:61-91, 124-139, 141-303
What differs, on purpose: the reference reads every loss with `.item()` right after each backward (a device sync per source per
step, Pretrain.py:79-91); here the loss tensors are queued and only read when a log line is due (`LossMeters.flush`), so the host
keeps enqueueing the next step while the GPU runs.  Region batches (run_region_iter) are outside the hot-path scope."""
import math
from collections import OrderedDict

import torch


class AttrDict(dict):
    """utils.AttrDict: the reference's configs are dicts read both as args['k'] and args.k."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight", "norm1.bias", "norm1.weight", "norm2.bias",
            "norm2.weight")  # optim.py:16-24 (substring match)


def optimizer_groups(model, init_params=None):
    """Names per group, in named_parameters order: [decay, no_decay, decay x lr_mult, no_decay x lr_mult] (optim.py:26-47)."""
    large_lr = set(init_params if init_params is not None else getattr(model, "init_params", []))
    groups = [[], [], [], []]
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        nd = any(t in n for t in NO_DECAY)
        groups[(2 if n in large_lr else 0) + (1 if nd else 0)].append(n)
    return groups


def create_optimizer(args, model):
    lr, wd = args.lr, args.weight_decay
    lr_mult = getattr(args, 'lr_mult', 1) if not isinstance(args, dict) else args.get('lr_mult', 1)
    params = dict(model.named_parameters())
    g = optimizer_groups(model)
    pg = [{"params": [params[n] for n in g[0]], "weight_decay": wd, "lr": lr},
          {"params": [params[n] for n in g[1]], "weight_decay": 0.0, "lr": lr},
          {"params": [params[n] for n in g[2]], "weight_decay": wd, "lr": lr * lr_mult},
          {"params": [params[n] for n in g[3]], "weight_decay": 0.0, "lr": lr * lr_mult}]
    # transformers' AdamW of the reference = decoupled weight decay applied after the Adam update, bias correction on, eps added to
    # sqrt(v) before the correction.  RCCLDDPAccelerator runs exactly that rule as one fused kernel over the flat arena
    # (csrc/elementwise.hip adamw_kernel); this torch.optim.AdamW object carries the groups / hyper-parameters / state_dict and is
    # the stepping rule only where the fused path is off (CPU): there eps enters after the correction (differs for |g| ~ 1e-8)
    return torch.optim.AdamW(pg, lr=lr, eps=1e-8, betas=(0.9, 0.98))


def linear_schedule(current_step, num_warmup_steps, num_training_steps):
    """scheduler.py:16-22."""
    if current_step < num_warmup_steps:
        return float(current_step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - current_step) / float(max(1, num_training_steps - num_warmup_steps)))


def create_scheduler(args, optimizer):
    if 'num_training_steps' not in args:
        args['num_training_steps'] = args['epochs'] * args['step_per_epoch']
    if isinstance(args['num_warmup_steps'], float):
        assert 0 <= args['num_warmup_steps'] < 1
        args['num_warmup_steps'] = int(args['num_training_steps'] * args['num_warmup_steps'])
    if args['sched'] != 'linear':
        raise NotImplementedError(f"args.sched == {args['sched']}")
    nw, nt = args['num_warmup_steps'], args['num_training_steps']
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda s: linear_schedule(s, nw, nt), last_epoch=-1)


class LossMeters:
    """Running means of the per-source losses without a device sync per update: tensors are parked and read in one go."""

    def __init__(self):
        self.pending = []
        self.total = OrderedDict()
        self.count = OrderedDict()

    def update(self, **kw):
        for k, v in kw.items():
            self.pending.append((k, v.detach() if torch.is_tensor(v) else v))

    def flush(self):
        for k, v in self.pending:
            self.total[k] = self.total.get(k, 0.0) + float(v)
            self.count[k] = self.count.get(k, 0) + 1
        self.pending = []

    def global_avg(self):
        self.flush()
        return {k: self.total[k] / self.count[k] for k in self.total}


_METER_NAMES = {'image': ('loss_itc', 'loss_itm', 'loss_mlm', 'loss_mim'), 'web': ('loss_witc', 'loss_witm', 'loss_wmlm', 'loss_wmim'),
                'imagenet': (None, None, None, 'loss_imim'), 'aux': (None, None, 'loss_amlm', None)}


def _to(device, t):
    return None if t is None else t.to(device, non_blocking=True)


def _backward(accelerator, loss, optimizer, last):
    """accelerator.backward_step; accelerators that take the `sync` hint (RCCLDDPAccelerator) exchange gradients only on the last
    backward before an optimizer step -- the sources of a multi-source step accumulate locally (the mean over ranks is linear)."""
    if getattr(accelerator, "takes_sync_hint", False):
        accelerator.backward_step(loss, optimizer, sync=last)
    else:
        accelerator.backward_step(loss, optimizer)


def run_image_iter(model, image_batch, optimizer, accelerator, metric_logger, device, data_source, ret_mim_loss=True,
                   ret_match_loss=True, ret_mlm_loss=True, ret_itc_loss=True, do_optm=False):
    """Pretrain.py:61-91."""
    image = _to(device, image_batch[0])
    extra = {}
    base = model.module if hasattr(model, 'module') else model
    if getattr(base, 'accepts_text_lens', False) and torch.is_tensor(image_batch[2]) and not image_batch[2].is_cuda:
        # caption lengths, read off the CPU batch before it is uploaded: the towers then run on unpadded token rows
        # (xfm_amd.packing) without ever asking the device for a size
        from .packing import lens_from_mask
        extra['text_lens'] = lens_from_mask(image_batch[2])
    text_ids, text_atts, text_ids_masked, masked_pos, masked_ids = (_to(device, t) for t in image_batch[1:])
    loss = model(image, text_ids, text_atts, text_ids_masked=text_ids_masked, masked_pos=masked_pos, masked_ids=masked_ids,
                 ret_match_loss=ret_match_loss, ret_mim_loss=ret_mim_loss, ret_mlm_loss=ret_mlm_loss, ret_itc_loss=ret_itc_loss,
                 data_source=data_source, **extra)
    _backward(accelerator, loss['loss_itc'] + loss['loss_itm'] + loss['loss_mlm'] + loss['loss_mim'], optimizer, do_optm)
    if do_optm:
        accelerator.optimizer_step(optimizer, model)
        optimizer.zero_grad()
    for key, name in zip(('loss_itc', 'loss_itm', 'loss_mlm', 'loss_mim'), _METER_NAMES[data_source]):
        if name is not None:
            metric_logger.update(**{name: loss[key]})


def run_text_iter(model, batch, optimizer, accelerator, metric_logger, device):
    """Pretrain.py:124-139: a text-only MLM step with its own optimizer step."""
    text_ids, text_atts, text_ids_masked, masked_pos, masked_ids = (_to(device, t) for t in batch)
    optimizer.zero_grad()
    loss = model(None, text_ids, text_atts, text_ids_masked=text_ids_masked, masked_pos=masked_pos, masked_ids=masked_ids)
    _backward(accelerator, loss['loss_mlm'], optimizer, True)
    accelerator.optimizer_step(optimizer, model)
    optimizer.zero_grad()
    metric_logger.update(loss_tmlm=loss['loss_mlm'])


def train(model, image_loader, data_loaders, optimizer, epoch_info, device, scheduler, config, accelerator, checkpointer=None,
          world_size=1, print_freq=50, log=None):
    """Pretrain.py:141-303.  data_loaders = (aux, web, imagenet, region, text) image/text sources, any of them None.
    Returns the averaged meters.  `checkpointer.save_checkpoint(model_state=..., epoch=..., step=..., training_states=...)` is
    called on the reference's schedule (ckpt_frequent epochs / ckpt_frequent_step steps) when given."""
    model.train()
    image_loader_aux, image_loader_web, image_loader_imagenet, region_loader, text_loader = data_loaders
    if region_loader is not None:
        raise NotImplementedError("region batches (run_region_iter, Pretrain.py:94-121) are outside the hot-path scope")
    start_epoch, max_epoch = epoch_info
    metric_logger = LossMeters()
    step_per_epoch = math.ceil(config['train_dataset_size'] / (config['batch_size'] * world_size))
    assert step_per_epoch > 1
    global_step = start_epoch * step_per_epoch + 1
    inf = float('inf')
    stop_itm, stop_mlm, stop_itc = config.get('stop_calc_itm', inf), config.get('stop_calc_mlm', inf), config.get('stop_calc_itc', inf)
    stop_mim, stop_mm = config.get('stop_calc_mim', inf), config.get('stop_calc_mm', inf)
    iters = {name: (iter(ld) if ld is not None else None)
             for name, ld in (('web', image_loader_web), ('imagenet', image_loader_imagenet), ('text', text_loader),
                              ('aux', image_loader_aux))}
    # utils.MetricLogger.log_every stops after (end_epoch - start_epoch) * step_per_epoch batches (Pretrain.py:205-209): a run resumed at
    # epoch k trains the REMAINING epochs, whatever the loader's length
    total_steps = (max_epoch - start_epoch) * step_per_epoch if max_epoch is not None else None
    for i, batch in enumerate(image_loader):
        if total_steps is not None and i >= total_steps:
            break
        gates = dict(ret_mim_loss=global_step < stop_mim, ret_match_loss=global_step < stop_itm, ret_mlm_loss=global_step < stop_mlm,
                     ret_itc_loss=global_step < stop_itc)
        if iters['text'] is not None:
            run_text_iter(model, next(iters['text']), optimizer, accelerator, metric_logger, device)
        if iters['web'] is not None:  # (the reference computes a do_optm here and does not pass it, Pretrain.py:225-227)
            run_image_iter(model, next(iters['web']), optimizer, accelerator, metric_logger, device, data_source='web', **gates)
        if iters['aux'] is not None:
            run_image_iter(model, next(iters['aux']), optimizer, accelerator, metric_logger, device, data_source='aux',
                           ret_mim_loss=gates['ret_mim_loss'], ret_match_loss=False, ret_mlm_loss=gates['ret_mlm_loss'],
                           ret_itc_loss=False)
        if iters['imagenet'] is not None:
            do_optm = not (global_step < stop_mm and iters['web'] is not None)
            run_image_iter(model, next(iters['imagenet']), optimizer, accelerator, metric_logger, device, data_source='imagenet',
                           do_optm=do_optm, **gates)
        if global_step < stop_mm:
            run_image_iter(model, batch, optimizer, accelerator, metric_logger, device, data_source='image', do_optm=True, **gates)
        metric_logger.update(lr=optimizer.param_groups[0]["lr"], lr_large=optimizer.param_groups[2]["lr"])
        scheduler.step()
        if (i + 1) % print_freq == 0:
            metric_logger.flush()
            if log is not None:
                log(global_step, metric_logger.global_avg())
        current_epoch = global_step // step_per_epoch
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        if checkpointer is not None and (not distributed or torch.distributed.get_rank() == 0):  # utils.is_main_process()
            at_epoch = (global_step + 1) % step_per_epoch == 0 and (current_epoch + 1) % config['ckpt_frequent'] == 0
            at_step = (global_step + 1) % config['ckpt_frequent_step'] == 0
            base = model.module if hasattr(model, 'module') else model
            if at_epoch:
                # optimizer.state_dict() carries the fused AdamW moments in torch's own per-parameter format
                # (RCCLDDPAccelerator._publish_optimizer_state), so the reference's save / resume code works unchanged
                opt_state = optimizer.state_dict()
                checkpointer.save_checkpoint(model_state={'model': base.state_dict(), 'optimizer': opt_state,
                                                          'lr_scheduler': scheduler.state_dict(), 'config': config,
                                                          'epoch': current_epoch},
                                             epoch=current_epoch, training_states=opt_state)
            if at_step:
                checkpointer.save_checkpoint(model_state={'model': base.state_dict(), 'config': config}, epoch=current_epoch,
                                             step=global_step, training_states=optimizer.state_dict())
        global_step += 1
    return {k: "{:.5f}".format(v) for k, v in metric_logger.global_avg().items()}


def resume(checkpoint, optimizer, lr_scheduler):
    """Pretrain.py:437-445 (`config['resume']`): optimizer + scheduler state and the epoch to continue from.  Called BEFORE
    accelerator.set_up like the reference does (set_up adopts the loaded AdamW moments, per-parameter step counts and live set
    into its arenas) or after it (the optimizer's load hook does the same)."""
    if isinstance(checkpoint, str):
        checkpoint = torch.load(checkpoint, map_location='cpu', weights_only=False)
    optimizer.load_state_dict(checkpoint['optimizer'])
    lr_scheduler.load_state_dict(checkpoint['lr_scheduler'])
    return checkpoint['epoch'] + 1
