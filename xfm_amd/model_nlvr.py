"""Drop-in for models/model_nlvr.py XFMForNLVR: two images per statement, each fused with the statement's text through the fusion
tower; the two [CLS] states are concatenated into a 2-way head."""
import torch
import torch.nn.functional as F

from .xfm import XFMBase, build_mlp


class XFMForNLVR(XFMBase):
    """model_nlvr.py:14-44."""

    def __init__(self, config):
        super().__init__(config, load_vision_params=False, load_text_params=False, use_contrastive_loss=False,
                         use_matching_loss=False, use_mlm_loss=False, use_bbox_loss=False)
        self.cls_head = build_mlp(input_dim=self.text_width * 2, output_dim=2)
        if 'load_domain_pretrained' not in config or not config['load_domain_pretrained']:
            self.init_params = ['cls_head.' + n for n, _ in self.cls_head.named_parameters()]

    def forward(self, image, text_ids, text_atts, targets, train=True):
        """image: [2B, 3, H, W] -- the B first images followed by the B second images (model_nlvr.py:28)."""
        image_embeds, image_atts = self.get_vision_embeds(image)
        encoder_embeds = self.get_text_embeds(text_ids, text_atts)
        n = targets.size(0)
        image0_embeds, image1_embeds = torch.split(image_embeds, n)
        cls0 = self.get_cross_embeds(image0_embeds, image_atts[:n], text_embeds=encoder_embeds, text_atts=text_atts,
                                     is_pretrain=False)[:, 0, :]
        cls1 = self.get_cross_embeds(image1_embeds, image_atts[n:], text_embeds=encoder_embeds, text_atts=text_atts,
                                     is_pretrain=False)[:, 0, :]
        output_cls = torch.cat((cls0, cls1), dim=-1)
        assert output_cls.shape[-1] == self.text_width * 2
        prediction = self.cls_head(output_cls)
        return F.cross_entropy(prediction.float(), targets) if train else prediction
