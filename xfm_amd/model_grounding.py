"""Drop-in for models/model_grounding.py XFMForGrounding (referring-expression grounding): the fused [CLS] of (image, expression)
regresses one box; L1 + generalised-IoU loss."""
from .xfm import XFMBase, load_pretrained


class XFMForGrounding(XFMBase):
    """model_grounding.py:36-62."""

    def __init__(self, config):
        super().__init__(config, load_vision_params=False, load_text_params=False, use_contrastive_loss=False,
                         use_matching_loss=False, use_mlm_loss=False, use_bbox_loss=True)
        self.init_params = []

    def load_pretrained(self, ckpt_rpath, config, load_bbox_pretrain=False, is_eval=False):
        state_dict = load_pretrained(self, ckpt_rpath, config, is_eval=is_eval, load_text=True)
        msg = self.load_state_dict(state_dict, strict=False)
        if self._arena is not None:
            self._arena.bump()
        return msg

    def forward(self, image, text_ids, text_atts, target_bbox=None):
        image_embeds, _ = self.get_vision_embeds(image)
        text_embeds = self.get_text_embeds(text_ids, text_atts)
        output_coord = self.predict_bbox(image_embeds, text_ids, text_atts, text_embeds, is_pretrain=False)
        if target_bbox is None:
            return output_coord
        loss_bbox, loss_giou = self.get_bbox_loss(output_coord, target_bbox)
        return output_coord, loss_bbox, loss_giou


class XFMForGroundingDomainPretrain(XFMBase):
    """model_grounding.py:12-33 needs the region path (`idx_to_group_img`: several boxes per image through one vision pass), which is
    outside the hot-path scope."""

    def __init__(self, config):
        raise NotImplementedError("XFMForGroundingDomainPretrain runs the region path (xfm.py:574-597, idx_to_group_img); use XFMForGrounding")
