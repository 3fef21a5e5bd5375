"""TextEncoder — drop-in for ref src/models/text_encoder.py:7-78 with the XLM-R forward on HIP.

`self.encoder` (HuggingFace AutoModel) is the parameter container only; `ser_xlmr_forward` runs the
embeddings and the 12 post-LN layers.  The tokenizer stays the HuggingFace one on the CPU, as in the
reference (:51).  The optional Whisper ASR branch (:39-49, :59-73) needs a second pretrained model
and is off by default in the reference; it is not part of this path.
"""
from typing import List, Optional

import torch
import torch.nn as nn

from .. import _lib as L
from .._engines import XlmrEngine
from .adapter import adapter_apply
from .pooling import AttentiveStatsPooling


class TextEncoder(nn.Module):
    def __init__(self, model_name="xlm-roberta-base", adapter_dim: int = 256, freeze_base: bool = True,
                 use_asr_integration: bool = False, asr_model_name: str = "openai/whisper-base",
                 precision: str = "bf16x3", hf_config=None, tokenizer=None):
        super().__init__()
        from transformers import AutoModel, AutoTokenizer
        if hf_config is not None:
            from transformers import XLMRobertaModel
            self.tokenizer = tokenizer
            self.encoder = XLMRobertaModel(hf_config)
        else:
            self.tokenizer = AutoTokenizer.from_pretrained(model_name)
            self.encoder = AutoModel.from_pretrained(model_name)
        self.freeze_base = freeze_base
        if freeze_base:
            for p in self.encoder.parameters():
                p.requires_grad = False
        hid = self.encoder.config.hidden_size
        self.adapter = nn.Sequential(nn.Linear(hid, adapter_dim), nn.ReLU(), nn.Linear(adapter_dim, hid))
        self.pool = AttentiveStatsPooling(hid)
        if use_asr_integration:
            raise NotImplementedError("the Whisper ASR branch is outside the HIP hot path (reference default: off)")
        self.use_asr_integration = False
        self.asr_integration = None
        self.asr_fusion = nn.Sequential(nn.Linear(hid + 8, hid), nn.ReLU(), nn.Dropout(0.1))
        self._asr_model_name = asr_model_name
        self.precision = precision
        self._engine = None
        from ._flat import FlatParams
        self._adapter_flat = FlatParams(list(self.adapter.parameters()))     # with the module, not on first use (see AudioEncoder)
        self._register_load_state_dict_pre_hook(lambda *a, **k: setattr(self, "_engine", None))

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def engine(self):
        dev = self.adapter[0].weight.device
        if self._engine is None or self._engine.device != dev:
            prec = L.PREC_BF16X3 if self.precision == "bf16x3" else L.PREC_BF16
            self._engine = XlmrEngine(self.encoder.config, self.encoder.state_dict(), dev, prec)
        return self._engine

    def forward_ids(self, input_ids: torch.Tensor, attention_mask: torch.Tensor):
        """Pre-tokenised entry: ids [B,S] int64, mask [B,S] -> (seq [B,S,H], mask float)."""
        dev = self.adapter[0].weight.device
        ids = input_ids.to(dev)
        mask = attention_mask.to(dev)
        noisy = self.training and getattr(self, "encoder_train_noise", False)
        if not self.freeze_base or noisy:
            # freeze_base=False: BASELINE config 3, every XLM-R parameter is trained (ref :13-15); encoder_train_noise: HF's
            # training-mode dropout sites, which the reference leaves on for the frozen encoder too (src/train.py:124)
            from ._finetune import Noise, xlmr_forward
            noise = None
            if noisy:
                if getattr(self, "_noise", None) is None:
                    self._noise = Noise(self.encoder.config, 1, seed=getattr(self, "noise_seed", 0))
                noise = self._noise
            from .. import _ops as O
            with O.linear_forward_products(1 if self.precision == "bf16" else 3):      # see AudioEncoder.encode
                if self.freeze_base:
                    with torch.no_grad():
                        seq = xlmr_forward(self.encoder, ids, mask, noise)
                else:
                    seq = xlmr_forward(self.encoder, ids, mask, noise)
            return adapter_apply(self, seq), mask.to(seq.dtype)
        with torch.no_grad():
            seq = self.engine().forward(ids, mask)
        return adapter_apply(self, seq), mask.to(seq.dtype)

    def forward(self, text_list: List[str], audio_waveforms=None):
        encoded = self.tokenizer(text_list, padding=True, truncation=True, return_tensors="pt")
        return self.forward_ids(encoded["input_ids"], encoded["attention_mask"])
