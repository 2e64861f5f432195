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
    """model_grounding.py:12-33: several (expression, box) samples per image through ONE vision pass -- `idx_to_group_img` [bs] names
    each sample's image (the region call form without region masks, xfm.py:577-588: every sample sees its whole image); box regression
    from the fused [CLS], L1 + GIoU with the `is_image` weighting."""

    def __init__(self, config):
        super().__init__(config, load_vision_params=False, load_text_params=False, use_contrastive_loss=False,
                         use_matching_loss=False, use_mlm_loss=False, use_bbox_loss=True)
        self.init_params = []

    def load_pretrained(self, ckpt_rpath, config):
        state_dict = load_pretrained(self, ckpt_rpath, config, is_eval=False, load_text=True)
        msg = self.load_state_dict(state_dict, strict=False)
        if self._arena is not None:
            self._arena.bump()
        return msg

    def forward(self, image, text_ids, text_atts, idx_to_group_img, target_bbox, is_image=None):
        image_embeds_fullatts, _ = self.get_vision_embeds(image, idx_to_group_img=idx_to_group_img)
        text_embeds = self.get_text_embeds(text_ids, text_atts)
        output_coord = self.predict_bbox(image_embeds_fullatts, text_ids, text_atts, text_embeds, is_pretrain=False)
        return self.get_bbox_loss(output_coord, target_bbox, is_image=is_image)
