"""oracle/ref_import.py -- AUTHORING-CONTAINER ONLY.  Imports the reference's own Python
(/root/reference, which does not exist on the GPU box) behind the stubs it needs on CPU
(SURVEY.md Appendix C).  Used by pin_against_reference.py and gen_golden.py; never by tests,
smoke() or bench.py at run time.
"""
import importlib.machinery
import os
import sys
import types

REF = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "model", "llm"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_done = {}


def import_x070():
    """model.llm.rwkv_s2s_single_ffn with deepspeed and the nvcc JIT stubbed out."""
    if "x070" in _done:
        return _done["x070"]
    for pth in (REF, os.path.join(REF, "third_party")):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    if "deepspeed" not in sys.modules:
        _stub("deepspeed", checkpointing=types.SimpleNamespace(checkpoint=lambda f, *a: f(*a)))
    import torch.utils.cpp_extension as ce
    ce.load = lambda *a, **k: None
    import model.llm.rwkv_s2s_single_ffn as ref
    _done["x070"] = ref
    return ref


def import_batch_twin():
    """model.llm.rwkv_asr_cuda_whisper (forward_batch stateful twin)."""
    if "batch" in _done:
        return _done["batch"]
    import_x070()
    if "torchaudio" not in sys.modules:
        _stub("torchaudio")
    import model.llm.rwkv_asr_cuda_whisper as ref_b
    _done["batch"] = ref_b
    return ref_b


def import_spark_layout():
    if "spark" in _done:
        return _done["spark"]
    import_x070()
    for n in ("sparktts", "sparktts.models", "soundfile"):
        if n not in sys.modules:
            _stub(n)
    _stub("sparktts.models.audio_tokenizer", BiCodecTokenizer=object)
    from inference.rwkv7speech_inference import create_inputs
    from data.utils import spark_dataset
    import utils.multiple_jsonl as mj
    _done["spark"] = types.SimpleNamespace(create_inputs=create_inputs, spark_dataset=spark_dataset,
                                           multiple_jsonl=mj)
    return _done["spark"]
