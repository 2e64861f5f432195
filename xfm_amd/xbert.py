"""BERT-flavoured text / fusion towers on the HIP hot path, behind the reference's `models/xbert.py` interface.

Same layer stack and kernels as `xfm_amd.xroberta` (state_dict keys of a layer are identical); what differs:
  * BertEmbeddings (xbert.py:168-215): absolute positions `position_ids[:, :T]`, token type row 0 of 2, word padding_idx 0,
    LayerNorm eps 1e-12  -> the embedding kernel runs with pos_mode=1;
  * BertSelfAttention (xbert.py:296-301, 329-330) scales the scores after QK^T unless `config.fp16`; the HIP attention
    kernel applies the scale to the fp32 scores in either case, so both orderings are served by the one kernel;
  * BertForMaskedLM (xbert.py:1523-1618): `.bert` is an attribute (not a method), one head `cls.predictions`
    (transform.dense -> GELU -> transform.LayerNorm -> decoder + shared bias, xbert.py:663-697);
  * BertLMHeadModel (xbert.py:1235-1347): the causal answer decoder of a bert-named VQA model (model_generation.py:54) -- `bert` +
    `cls`, causal self-attention mask, next-token shift, `reduction='none'` CE summed per sequence.
"""
import json
from types import SimpleNamespace

import torch
import torch.nn as nn

from .arena import LinearSlot, OwnsArena, ParamArena
from .beit2 import _Affine
from .ops import lm_head_ce, lm_head_logits
from .xroberta import RobertaModel, _Emb, _Lin


class BertConfig:
    """The subset of transformers' BertConfig the path reads (xfm.py:275-284)."""

    def __init__(self, **kw):
        d = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                 hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=512,
                 type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12, pad_token_id=0, fusion_layer=12,
                 encoder_width=768, add_cross_attention=False, fp16=False)
        d.update(kw)
        self.__dict__.update(d)

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))


class BertEmbeddings(nn.Module):
    pos_mode = 1  # absolute positions 0..T-1

    def __init__(self, config):
        super().__init__()
        std = config.initializer_range
        self.word_embeddings = _Emb(config.vocab_size, config.hidden_size, std, config.pad_token_id)
        self.position_embeddings = _Emb(config.max_position_embeddings, config.hidden_size, std)
        self.token_type_embeddings = _Emb(config.type_vocab_size, config.hidden_size, std)
        self.LayerNorm = _Affine(config.hidden_size, config.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.padding_idx = config.pad_token_id
        self.p_drop = config.hidden_dropout_prob


class BertModel(RobertaModel):
    """xbert.py:845-1130.  Forward keywords, `mode` layer ranges and the returned namespace are RobertaModel's."""
    embeddings_class = BertEmbeddings


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.hidden_act != "gelu":
            raise NotImplementedError("only the exact-erf GELU head activation is implemented")
        self.dense = _Lin(config.hidden_size, config.hidden_size, config.initializer_range)
        self.LayerNorm = _Affine(config.hidden_size, config.layer_norm_eps)


class BertLMPredictionHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = _Lin(config.hidden_size, config.vocab_size, config.initializer_range)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.decoder.bias = self.bias  # tied, xbert.py:690-691

    @property
    def layer_norm(self):  # the name the fused LM-head node uses
        return self.transform.LayerNorm

    def linear_slots(self, prefix):
        self._slot_dense = LinearSlot(prefix + "transform.dense", [self.transform.dense.weight], [self.transform.dense.bias])
        self._slot_decoder = LinearSlot(prefix + "decoder", [self.decoder.weight], [self.bias], pad_k_to=64)
        return [self._slot_dense, self._slot_decoder]


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)


class BertForMaskedLM(OwnsArena, nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.bert = BertModel(config, add_pooling_layer=False)
        self.cls = BertOnlyMLMHead(config)
        self._arena = None

    def linear_slots(self, prefix=""):
        return self.bert.linear_slots(prefix + "bert.") + self.cls.predictions.linear_slots(prefix + "cls.predictions.")

    def attach(self, arena):
        self._arena = arena
        self.bert.attach(arena)

    def finalize(self, device=None):
        device = device or self.cls.predictions.bias.device
        self.attach(ParamArena(self, self.linear_slots(), device))
        self._own_arena = True
        return self

    def gather_seq_out_by_pos(self, seq, pos):
        return torch.gather(seq, 1, pos.unsqueeze(2).expand(-1, -1, seq.size(-1)))

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None,
                labels=None, output_attentions=None, output_hidden_states=None, return_dict=None, is_decoder=False,
                mode='multi_modal', return_logits=False, masked_pos=None):
        outputs = self.bert(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids,
                            position_ids=position_ids, head_mask=head_mask, inputs_embeds=inputs_embeds,
                            encoder_embeds=encoder_embeds, encoder_hidden_states=encoder_hidden_states,
                            encoder_attention_mask=encoder_attention_mask, is_decoder=is_decoder, mode=mode)
        seq = outputs.last_hidden_state
        if masked_pos is not None:
            seq = self.gather_seq_out_by_pos(seq, masked_pos)
        head = self.cls.predictions
        V = self.config.vocab_size
        Bq, Tq = seq.shape[:2]
        if return_logits or labels is None:
            logits = lm_head_logits(seq.reshape(-1, seq.shape[-1]), head).view(Bq, Tq, V)
            if return_logits:
                return logits
            return SimpleNamespace(loss=None, logits=logits, hidden_states=None, attentions=None)
        loss, logits = lm_head_ce(seq.reshape(-1, seq.shape[-1]), head, labels.reshape(-1), 'mean')
        return SimpleNamespace(loss=loss, logits=logits[:, :V].view(Bq, Tq, V), hidden_states=None, attentions=None)


class BertLMHeadModel(OwnsArena, nn.Module):
    """Causal decoder with cross-attention to encoder states on the BERT-flavoured stack (xbert.py:1235-1347): what
    model_generation.py:54 builds as `text_decoder` for a bert-named text encoder.  Same fused stack and LM-head node as
    xroberta.RobertaForCausalLM; the state_dict keys are the reference's (`bert.*`, `cls.predictions.*`).  Label smoothing
    (xbert.py:1336-1337, LabelSmoothSoftmaxCEV1) is used by no XFM task model: > 0 raises.  Generation (beam search, past_key_values)
    is outside the hot-path scope, as for the RoBERTa decoder."""

    def __init__(self, config, label_smoothing=0.0):
        super().__init__()
        if label_smoothing > 0:
            raise NotImplementedError("label smoothing (xbert.py:1336-1337) is not used on the XFM path")
        self.config = config
        self.bert = BertModel(config, add_pooling_layer=False)
        self.cls = BertOnlyMLMHead(config)
        self.label_smoothing = label_smoothing
        self._arena = None

    def linear_slots(self, prefix=""):
        return self.bert.linear_slots(prefix + "bert.") + self.cls.predictions.linear_slots(prefix + "cls.predictions.")

    def attach(self, arena):
        self._arena = arena
        self.bert.attach(arena)

    def finalize(self, device=None):
        device = device or self.cls.predictions.bias.device
        self.attach(ParamArena(self, self.linear_slots(), device))
        self._own_arena = True
        return self

    def get_output_embeddings(self):
        return self.cls.predictions.decoder

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, labels=None, past_key_values=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, is_decoder=True,
                reduction='mean', mode='multi_modal', return_logits=False):
        if past_key_values is not None or use_cache:
            raise NotImplementedError("incremental decoding caches (generation) are outside the hot-path scope")
        outputs = self.bert(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, position_ids=position_ids,
                            head_mask=head_mask, inputs_embeds=inputs_embeds, encoder_hidden_states=encoder_hidden_states,
                            encoder_attention_mask=encoder_attention_mask, is_decoder=is_decoder, mode=mode)
        seq = outputs.last_hidden_state
        head = self.cls.predictions
        V = self.config.vocab_size
        B, T = seq.shape[:2]
        if return_logits or labels is None:
            logits = lm_head_logits(seq.reshape(-1, seq.shape[-1]), head).view(B, T, V)
            if return_logits:
                return logits[:, :-1, :].contiguous()
            return SimpleNamespace(loss=None, logits=logits, hidden_states=seq, past_key_values=None, attentions=None,
                                   cross_attentions=None)
        shifted, lab = seq[:, :-1, :], labels[:, 1:]   # next-token prediction (xbert.py:1331-1333)
        loss, logits = lm_head_ce(shifted.reshape(-1, seq.shape[-1]), head, lab.reshape(-1), reduction)
        if reduction == 'none':
            loss = loss.view(B, -1).sum(1)
        return SimpleNamespace(loss=loss, logits=logits[:, :V].view(B, T - 1, V), hidden_states=seq, past_key_values=None,
                               attentions=None, cross_attentions=None)
