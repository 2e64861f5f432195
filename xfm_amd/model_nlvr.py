"""Drop-in for models/model_nlvr.py XFMForNLVR: two images per statement, each fused with the statement's text through the fusion
tower; the two [CLS] states are concatenated into a 2-way head."""
import torch
import torch.nn.functional as F

from .ops import small_ce
from .xfm import XFMBase, build_mlp
from .xroberta import rowwise


class XFMForNLVR(XFMBase):
    """model_nlvr.py:14-44."""

    def __init__(self, config):
        super().__init__(config, load_vision_params=False, load_text_params=False, use_contrastive_loss=False,
                         use_matching_loss=False, use_mlm_loss=False, use_bbox_loss=False)
        self.cls_head = build_mlp(input_dim=self.text_width * 2, output_dim=2)
        if 'load_domain_pretrained' not in config or not config['load_domain_pretrained']:
            self.init_params = ['cls_head.' + n for n, _ in self.cls_head.named_parameters()]

    def forward(self, image, text_ids, text_atts, targets, train=True):
        """image: [2B, 3, H, W] -- the B first images followed by the B second images (model_nlvr.py:28).
        The reference fuses the statement with each image in two B-row passes (:31-35); here both pairs go through the fusion tower as
        ONE 2B-row pass (statement rows repeated), which computes the same rows with half the launches."""
        image_embeds, image_atts = self.get_vision_embeds(image)
        statement = self.get_text_embeds(text_ids, text_atts)
        n = targets.size(0)
        assert image_embeds.size(0) == 2 * n, "two images per statement"
        fused_cls = self.get_cross_embeds(image_embeds, image_atts, text_embeds=rowwise(lambda t: t.repeat(2, 1, 1), statement), text_atts=text_atts.repeat(2, 1),
                                          is_pretrain=False)[:, 0, :]
        pair = torch.cat((fused_cls[:n], fused_cls[n:]), dim=-1)   # [B, 2 * width]: (first image | second image)
        assert pair.shape[-1] == self.text_width * 2
        logits = self.cls_head(pair)
        return small_ce(logits, targets) if train else logits
