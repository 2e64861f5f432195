"""Bounding-box helpers of models/box_ops.py (cx, cy, w, h <-> x0, y0, x1, y1; IoU; generalised IoU), elementwise over PAIRED boxes:
the reference builds the full N x M matrix and takes its diagonal (xfm.py:831); the same numbers without the N^2 work."""
import torch


def box_cxcywh_to_xyxy(x):
    """box_ops.py:9-13."""
    x_c, y_c, w, h = x.unbind(-1)
    return torch.stack([x_c - 0.5 * w, y_c - 0.5 * h, x_c + 0.5 * w, y_c + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(x):
    """box_ops.py:16-20."""
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def box_area(b):
    return (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])


def paired_box_iou(boxes1, boxes2):
    """diag of box_ops.box_iou (box_ops.py:24-37): (iou, union) of boxes1[i] with boxes2[i]."""
    lt = torch.max(boxes1[:, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = box_area(boxes1) + box_area(boxes2) - inter
    return inter / union, union


def paired_generalized_box_iou(boxes1, boxes2):
    """diag of box_ops.generalized_box_iou (box_ops.py:40-59)."""
    iou, union = paired_box_iou(boxes1, boxes2)
    lt = torch.min(boxes1[:, :2], boxes2[:, :2])
    rb = torch.max(boxes1[:, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, 0] * wh[:, 1]
    return iou - (area - union) / area
