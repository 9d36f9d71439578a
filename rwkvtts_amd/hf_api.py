"""The slice of the transformers PreTrainedModel surface that the reference's scripts touch
(train_scripts/train_*.py, inference/*.py, data/utils/*.py): .device/.dtype, get/set_*_embeddings,
gradient_checkpointing_enable, save_pretrained/from_pretrained (config.json + model.safetensors).
The reference inherits it from transformers==4.51.3 (requirements.txt:245); the image carries 5.x, so the
heads here are plain nn.Modules that implement this surface themselves."""
import json
import os

import torch


class HFModelMixin:
    config_class = None

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def get_input_embeddings(self):
        return self.model.embeddings

    def set_input_embeddings(self, value):
        self.model.embeddings = value

    def get_output_embeddings(self):
        return getattr(self, "lm_head", None)

    def set_output_embeddings(self, new_embeddings):
        self.lm_head = new_embeddings

    def get_decoder(self):
        return self.model

    def set_decoder(self, decoder):
        self.model = decoder

    def gradient_checkpointing_enable(self, *a, **k):
        self.model.gradient_checkpointing = True

    def gradient_checkpointing_disable(self):
        self.model.gradient_checkpointing = False

    def init_weights(self, seed=0):
        from .backbone import init_weights
        init_weights(self, self.config, seed)
        return self

    def save_pretrained(self, path):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.config.to_dict(), f, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "model.safetensors"))

    @classmethod
    def from_pretrained(cls, path, torch_dtype=None, device=None, **unused):
        from safetensors.torch import load_file
        with open(os.path.join(path, "config.json")) as f:
            cfg = cls.config_class.from_dict(json.load(f))
        model = cls(cfg)
        model.load_state_dict(load_file(os.path.join(path, "model.safetensors")), strict=True)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        if device is not None:
            model = model.to(device)
        return model
