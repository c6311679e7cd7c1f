"""reference import path llamagen.language.t5 (test_llamagen.py:18, model_loader.py load_llamagen).

The caption encoder is conditioning PRE-processing (out of scope for the SJD hot path): this class keeps the
reference's constructor / `get_text_embeddings(texts) -> (embeddings [B, L, 2048], mask [B, L])` contract on top of transformers'
T5EncoderModel loaded from a LOCAL directory (there is no network on the GPU box), and `T5Embedder.from_embeddings` lets a driver
feed pre-computed caption embeddings through the same interface."""
import os

import torch


class T5Embedder:
    available_models = ['t5-v1_1-xxl', 't5-v1_1-xl', 'flan-t5-xl']

    def __init__(self, device, dir_or_name='t5-v1_1-xxl', *, local_cache=False, cache_dir=None, hf_token=None, use_text_preprocessing=True,
                 t5_model_kwargs=None, torch_dtype=None, use_offload_folder=None, model_max_length=120):
        self.device = torch.device(device)
        self.torch_dtype = torch_dtype or torch.bfloat16
        self.model_max_length = model_max_length
        self.use_text_preprocessing = use_text_preprocessing
        path = os.path.join(cache_dir or os.path.expanduser('~/.cache/IF_'), dir_or_name) if local_cache else dir_or_name
        if not os.path.isdir(path):
            raise FileNotFoundError(f"T5 checkpoint directory {path!r} not found (no hub access here); place the files there or use "
                                    "T5Embedder.from_embeddings(...)")
        from transformers import AutoTokenizer, T5EncoderModel
        self.tokenizer = AutoTokenizer.from_pretrained(path)
        self.model = T5EncoderModel.from_pretrained(path, torch_dtype=self.torch_dtype).to(self.device).eval()
        self._fixed = None

    @classmethod
    def from_embeddings(cls, embeddings, masks):
        """embeddings [B, L, C], masks [B, L] (1 = valid): a stand-in whose get_text_embeddings returns them as is."""
        self = cls.__new__(cls)
        self.device, self.torch_dtype, self.model_max_length = embeddings.device, embeddings.dtype, embeddings.shape[1]
        self.tokenizer = self.model = None
        self._fixed = (embeddings, masks)
        return self

    @staticmethod
    def text_preprocessing(text):
        return " ".join(str(text).lower().strip().split())

    @torch.no_grad()
    def get_text_embeddings(self, texts):
        if self._fixed is not None:
            return self._fixed
        if self.use_text_preprocessing:
            texts = [self.text_preprocessing(t) for t in texts]
        tok = self.tokenizer(texts, max_length=self.model_max_length, padding='max_length', truncation=True, return_attention_mask=True,
                             add_special_tokens=True, return_tensors='pt')
        ids, mask = tok['input_ids'].to(self.device), tok['attention_mask'].to(self.device)
        emb = self.model(input_ids=ids, attention_mask=mask)['last_hidden_state'].detach()
        return emb, mask
