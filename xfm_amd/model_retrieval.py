"""Drop-in for models/model_retrieval.py (image-text retrieval fine-tuning): ITC with `idx` soft labels + ITM with the text
tower's gradient kept (`is_pretrain=False`), on the HIP towers."""
from .xfm import XFMBase


class XFMForRetrieval(XFMBase):
    """model_retrieval.py:11-36."""

    def __init__(self, config):
        super().__init__(config, load_vision_params=False, load_text_params=False, use_contrastive_loss=True,
                         use_matching_loss=True, use_mlm_loss=False, use_bbox_loss=False)
        self.num_attention_heads = self.text_encoder.config.num_attention_heads
        self.init_params = []

    def load_pretrained(self, ckpt_rpath, config, is_eval=False):
        """model_retrieval.py:19-24: a pre-training checkpoint into the bare-encoder fine-tuning model."""
        from .xfm import load_pretrained
        state_dict = load_pretrained(self, ckpt_rpath, config, is_eval=is_eval, load_text=True)
        msg = self.load_state_dict(state_dict, strict=False)
        if self._arena is not None:
            self._arena.bump()
        return msg

    def forward(self, image, text_ids, text_atts, idx=None, neg_idx=None):
        from .model_pretrain import towers_side_by_side
        image_embeds, image_atts, text_embeds = towers_side_by_side(self, image, text_ids, text_atts)
        image_feat, text_feat = self.get_features(image_embeds, text_embeds)
        loss_itc = self.get_contrastive_loss(image_feat, text_feat, idx=idx)
        loss_itm = self.get_matching_loss(image_embeds, image_atts, image_feat, text_ids, text_atts, text_feat, idx=idx,
                                          text_embeds=text_embeds, is_pretrain=False, neg_idx=neg_idx)
        return loss_itc, loss_itm
