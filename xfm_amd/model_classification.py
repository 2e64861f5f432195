"""Drop-in for models/model_classification.py XFMForClassification: ImageNet (vision tower only, cls + mean-patch features through the
deep MLP head), text-only GLUE-style classification on the text tower, and the multimodal branch through the fusion tower."""
import torch
import torch.nn.functional as F

from .ops import small_ce
from .xfm import XFMBase, _DeepMlp, build_mlp


class XFMForClassification(XFMBase):
    """model_classification.py:10-93."""

    def __init__(self, config, init_scale=0.001):
        super().__init__(config, load_vision_params=True, load_text_params=False, use_contrastive_loss=False,
                         use_matching_loss=False, use_mlm_loss=False, use_bbox_loss=False)
        feature_dim = self.vision_width if config.get('task_name') == 'imagenet' else self.text_width
        self.is_lp = config.get('is_lp', False)
        if config.get('use_ofa', False):
            raise NotImplementedError("the OFA linear-probe encoder (model_classification.py:60-66) is outside the hot-path scope")
        task_name = config.get('task_name', 'glue')
        if task_name == 'imagenet' or self.is_lp:
            self.cls_head = self.build_mlp(input_dim=feature_dim * 2, output_dim=config['num_labels'])
        else:
            self.cls_head = build_mlp(input_dim=feature_dim, output_dim=config['num_labels'])
        self.init_params = ['cls_head.' + n for n, _ in self.cls_head.named_parameters()]

    def build_mlp(self, input_dim, output_dim):
        return _DeepMlp(input_dim, output_dim)

    def forward(self, image, text_ids, text_atts, targets, train=True):
        if image is None:
            output_cls = self.get_text_embeds(text_ids, text_atts)[:, 0, :]  # the bare text encoder (model_classification.py:52-55)
        elif text_ids is None:
            with torch.set_grad_enabled(torch.is_grad_enabled() and not self.is_lp):
                image_embeds, _ = self.get_vision_embeds(image)
            output_cls = torch.cat([image_embeds[:, 0, :], torch.mean(image_embeds[:, 1:, :].float(), dim=1).to(image_embeds.dtype)], dim=-1)
        else:
            image_embeds, image_atts = self.get_vision_embeds(image)
            encoder_embeds = self.get_text_embeds(text_ids, text_atts)
            output_cls = self.get_cross_embeds(image_embeds, image_atts, text_embeds=encoder_embeds, text_atts=text_atts,
                                               is_pretrain=False)[:, 0, :]
        prediction = self.cls_head(output_cls)
        if prediction.shape[-1] == 1:
            loss = F.mse_loss(prediction.view(-1).float(), targets.view(-1).float())
            return loss if train else prediction
        return small_ce(prediction, targets) if train else prediction
