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

    def save_pretrained(self, path, **unused):
        """config.json (+ `auto_map` and the modeling_rwkvspeech.py shim for the Spark model, as the reference's checkpoints
        carry them: model/test/audio_rwkv.config:9-13) + model.safetensors with rwkvfla keys."""
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        cfg = self.config.to_dict()
        if type(self).__name__ == "RWKV7ForSpeech":
            from . import modeling_rwkvspeech as shim
            cfg["auto_map"] = dict(shim.AUTO_MAP)
            with open(os.path.join(path, "modeling_rwkvspeech.py"), "w") as f:
                f.write(shim.SHIM_SOURCE)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(path, "model.safetensors"))

    @classmethod
    def register_for_auto_class(cls, auto_class="AutoModel"):
        """transformers' Auto* loaders call this on classes they reach through `auto_map`; nothing to record here."""

    @classmethod
    def from_pretrained(cls, path, *model_args, config=None, torch_dtype=None, dtype=None, device=None, device_map=None, **unused):
        """Also the entry transformers' AutoModelForCausalLM.from_pretrained(dir, trust_remote_code=True) lands in (through
        config.json's auto_map -> modeling_rwkvspeech.py): `config` is then the object AutoConfig built from the same file."""
        from safetensors.torch import load_file
        if config is not None and isinstance(config, cls.config_class):
            cfg = config
        else:
            with open(os.path.join(path, "config.json")) as f:
                cfg = cls.config_class.from_dict(json.load(f))
        torch_dtype = torch_dtype if torch_dtype is not None else dtype
        if isinstance(torch_dtype, str):
            torch_dtype = getattr(torch, torch_dtype) if torch_dtype != "auto" else None
        if device is None and isinstance(device_map, (str, torch.device)) and str(device_map) != "auto":
            device = device_map
        model = cls(cfg)
        model.load_state_dict(load_file(os.path.join(path, "model.safetensors")), strict=True)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        if device is not None:
            model = model.to(device)
        return model
