"""Drop-in for models/model_generation.py XFMForVQA (BASELINE configs[3]): the question goes through the text tower and the fusion
tower (cross-attention to the image), the answers through a causal decoder that cross-attends to the fused question states.
Training: per-answer sequence loss weighted by the annotators' answer weights.  Inference: answers are RANKED, not generated --
first-token probabilities pick k candidates, their full-sequence log-likelihood re-ranks them (rank_answer)."""
import copy
import os
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from .xfm import RobertaConfig, XFMBase, load_pretrained
from .xroberta import RobertaForCausalLM


def tile(x, dim, n_tile):
    """Each entry of `x` along `dim` repeated n_tile times, kept adjacent (model_generation.py:382-388)."""
    return x.repeat_interleave(n_tile, dim=dim)


def _fields(t):
    """The reference passes tokenizer outputs (`.input_ids`, `.attention_mask`); (ids, mask) pairs and dicts are accepted too."""
    if isinstance(t, (tuple, list)):
        return SimpleNamespace(input_ids=t[0], attention_mask=t[1])
    if isinstance(t, dict):
        return SimpleNamespace(input_ids=t["input_ids"], attention_mask=t["attention_mask"])
    return t


class XFMForVQA(XFMBase):
    """model_generation.py:23-202."""

    def __init__(self, config):
        super().__init__(config, load_vision_params=False, load_text_params=False, use_contrastive_loss=False,
                         use_matching_loss=False, use_mlm_loss=False, use_bbox_loss=False)
        assert isinstance(config['pad_token_id'], int)
        self.pad_token_id = config['pad_token_id']
        config_enc = self.text_encoder.config
        roberta = 'roberta' in config['text_encoder']
        if roberta:
            cfg_cls, dec_cls = RobertaConfig, RobertaForCausalLM
        else:   # bert-named text encoders decode with xbert.BertLMHeadModel (model_generation.py:52-54)
            from .xbert import BertConfig, BertLMHeadModel
            cfg_cls, dec_cls = BertConfig, BertLMHeadModel
        if 'text_config' in config:
            config_dec = cfg_cls(**config['text_config'])
        elif os.path.exists(os.path.join(config['text_encoder'], 'config.json')):
            config_dec = cfg_cls.from_json_file(os.path.join(config['text_encoder'], 'config.json'))
        else:
            config_dec = copy.deepcopy(config_enc)
        config_dec.encoder_width = config_enc.hidden_size
        config_dec.fusion_layer = config['decoder_fusion_start_at']  # first decoder layer with cross-attention
        config_dec.num_hidden_layers = config['num_dec_layers']
        self.cross_encoder_width = config_enc.encoder_width  # = vision width
        self.dec_encoder_width = config_enc.hidden_size
        self.text_decoder = dec_cls(config=config_dec)
        if self.dec_encoder_width != self.cross_encoder_width:
            self.init_params = ['text_decoder.' + n for n, _ in self.text_decoder.named_parameters()
                                if ('crossattention.self.key' in n) or ('crossattention.self.value' in n)]
        else:
            self.init_params = []

    def load_pretrained(self, ckpt_rpath, config, is_eval=False):
        """model_generation.py:61-91: a pre-training checkpoint into the VQA model -- text tower keys lose their `roberta.` level and
        the decoder starts as a copy of the fusion tower."""
        if is_eval:
            state_dict = load_pretrained(self, ckpt_rpath, config, is_eval=True)
        else:
            state_dict = load_pretrained(self, ckpt_rpath, config, load_text=False)
            for key in list(state_dict.keys()):
                name_to_replace = 'roberta.' if 'roberta' in config['text_encoder'] else 'bert.'
                if name_to_replace in key and 'text_encoder' in key:
                    state_dict[key.replace(name_to_replace, '')] = state_dict[key]
                    del state_dict[key]
                if 'fusion_encoder.' in key:
                    state_dict[key.replace('fusion_encoder', 'text_decoder')] = state_dict[key]
        msg = self.load_state_dict(state_dict, strict=False)
        if self._arena is not None:
            self._arena.bump()
        return msg

    def _question_states(self, image, question):
        from .model_pretrain import towers_side_by_side
        image_embeds, image_atts, text_embeds = towers_side_by_side(self, image, question.input_ids, question.attention_mask)
        return self.get_cross_embeds(image_embeds, image_atts, text_embeds=text_embeds, text_atts=question.attention_mask,
                                     is_pretrain=False)

    def forward(self, image, quesiton, answer=None, k=None, weights=None, train=True):
        """(the reference spells the argument `quesiton`; kept for keyword compatibility.)
        train: k[b] = number of answers of question b, weights = one weight per answer; returns the weighted loss / batch size.
        eval: answer = the candidate answer list, k = how many to re-rank; returns (topk_ids, topk_probs)."""
        question, answer = _fields(quesiton), _fields(answer)
        question_output = self._question_states(image, question)
        if train:
            answer_targets = answer.input_ids.masked_fill(answer.input_ids == self.pad_token_id, -100)
            # each question once per answer (the reference stacks Python lists, :112-117); the row index is built on the host from
            # the host-side counts, so nothing waits for the device
            rows = torch.tensor([b for b, n in enumerate(k) for _ in range(int(n))], device=question_output.device)
            question_states = question_output.index_select(0, rows)
            question_atts = question.attention_mask.index_select(0, rows)
            answer_output = self.text_decoder(answer.input_ids, attention_mask=answer.attention_mask,
                                              encoder_hidden_states=question_states, encoder_attention_mask=question_atts,
                                              labels=answer_targets, return_dict=True, reduction='none')
            loss = weights * answer_output.loss
            return loss.sum() / image.size(0)
        question_atts = torch.ones(question_output.size()[:-1], dtype=torch.long, device=question_output.device)
        return self.rank_answer(question_output, question_atts, answer.input_ids, answer.attention_mask, k)

    def rank_answer(self, question_states, question_atts, answer_ids, answer_atts, k):
        """model_generation.py:146-202."""
        num_ques = question_states.size(0)
        start_ids = answer_ids[0, 0].repeat(num_ques, 1)  # bos token
        start_output = self.text_decoder(start_ids, encoder_hidden_states=question_states, encoder_attention_mask=question_atts,
                                         return_dict=True, reduction='none')
        logits = start_output.logits[:, 0, :].float()  # first token's logits
        answer_first_token = answer_ids[:, 1]
        prob_first_token = F.softmax(logits, dim=1).index_select(dim=1, index=answer_first_token)
        topk_probs, topk_ids = prob_first_token.topk(k, dim=1)
        flat = topk_ids.reshape(-1)                        # [num_ques * k] candidate rows, question-major
        input_ids = answer_ids.index_select(0, flat)
        input_atts = answer_atts.index_select(0, flat)
        targets_ids = input_ids.masked_fill(input_ids == self.pad_token_id, -100)
        question_states = tile(question_states, 0, k)
        question_atts = tile(question_atts, 0, k)
        output = self.text_decoder(input_ids, attention_mask=input_atts, encoder_hidden_states=question_states,
                                   encoder_attention_mask=question_atts, labels=targets_ids, return_dict=True, reduction='none')
        answer_loss = output.loss.view(input_ids.size(0), -1)
        # chain rule: log p(first token) + log p(rest | first)
        log_probs = torch.cat([topk_probs.view(-1, 1).log(), -answer_loss], dim=1)
        log_probs_sum = log_probs.sum(1).view(num_ques, k)
        topk_probs = F.softmax(log_probs_sum, dim=-1)
        topk_probs, rerank_id = topk_probs.topk(k, dim=1)
        topk_ids = torch.gather(topk_ids, 1, rerank_id)
        return topk_ids, topk_probs
