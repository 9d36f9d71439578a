"""The `auto_map` target of the reference's checkpoints (model/test/audio_rwkv.config:9-13 names
`modeling_rwkvspeech.RWKV7SpeechConfig / RWKV7Model / RWKV7ForSpeech`; data/spark/modeling_rwkvspeech.py:1-6 is the module the
reference ships next to config.json), served by the HIP classes -- so that

    AutoModelForCausalLM.from_pretrained(ckpt_dir, trust_remote_code=True)      (train_spark_rwkv7speech.py, inference/*.py)

hands back rwkvtts_amd.spark_llm.RWKV7ForSpeech.  save_pretrained() writes a two-line `modeling_rwkvspeech.py` importing this
module next to config.json; for an existing reference checkpoint, drop that file over the reference's one."""
from .backbone import RWKV7Model
from .spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig

RWKV7ForCausalLM = RWKV7ForSpeech
RWKV7Config = RWKV7SpeechConfig

__all__ = ["RWKV7ForSpeech", "RWKV7SpeechConfig", "RWKV7Model", "RWKV7ForCausalLM", "RWKV7Config"]

SHIM_SOURCE = '''"""auto_map shim: the classes named in config.json, served by the MI355X-native package (rwkvtts_amd)."""
from rwkvtts_amd.modeling_rwkvspeech import RWKV7Config, RWKV7ForCausalLM, RWKV7ForSpeech, RWKV7Model, RWKV7SpeechConfig  # noqa: F401
'''
AUTO_MAP = {"AutoConfig": "modeling_rwkvspeech.RWKV7SpeechConfig", "AutoModel": "modeling_rwkvspeech.RWKV7Model",
            "AutoModelForCausalLM": "modeling_rwkvspeech.RWKV7ForSpeech"}
