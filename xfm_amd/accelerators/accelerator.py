class Accelerator:
    """The 3-method contract of the reference (accelerators/accelerator.py:15-32)."""

    def __init__(self, cfg, logger=None):
        self.cfg = cfg
        self.logger = logger

    def set_up(self, model, optimizer, lr_scheduler, local_rank, world_size, rank):
        raise NotImplementedError("Set Up method not implement in Accelerator, please check! ")

    def broadcast(self):
        raise NotImplementedError("Broadcast method not implement in Accelerator, please check! ")

    def backward_step(self, loss, optimizer):
        loss.backward()

    def optimizer_step(self, optimizer, model, grad_norm: float = 0.0) -> float:
        raise NotImplementedError

    def get_metrics(self):
        return {}

    def state_dict(self):
        return {}

    def load_state_dict(self, state_dict):
        pass
