"""Pins rwkvtts_amd/trainer.py's optimizer groups and cosine schedule against the reference's own functions
(authoring container only: /root/reference does not exist on the GPU box).

train_scripts/train_cosy_rwkv7speech_multiple_dataset.py is a script (argparse, deepspeed, wandb at import), so the two
functions are cut out of its source by name (ast) and executed here:
  configure_optimizer (:162-202)      on the named_parameters of a small RWKV7CosyLM built from OUR package (rwkvfla key
                                      names), with deepspeed.ops.adam.FusedAdam replaced by a recorder of optim_groups;
  update_learning_rate (:224-244)     on a duck-typed optimizer with the three param groups.
Only the resulting DATA is committed: tests/golden/optimizer_groups.npz (parameter name -> group, lr scale, weight decay)
and tests/golden/lr_schedule.npz (lr per group at a list of steps).        python oracle/pin_optimizer.py [--write]
"""
import ast
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference/train_scripts/train_cosy_rwkv7speech_multiple_dataset.py"
GOLD = os.path.join(REPO, "tests", "golden")


def ref_functions():
    src = open(REF).read()
    tree = ast.parse(src)
    want = {"configure_optimizer", "update_learning_rate"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert {n.name for n in body} == want
    ns = {}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    return ns["configure_optimizer"], ns["update_learning_rate"]


class _Recorder:
    def __init__(self, groups, **kw):
        self.param_groups = groups
        self.kw = kw


def small_cosy_model():
    from rwkvtts_amd.cosy_llm import RWKV7CosyConfig, RWKV7CosyLM
    cfg = RWKV7CosyConfig(vocab_size=50, speech_token_size=20, hidden_size=128, num_hidden_layers=2, decay_low_rank_dim=32,
                          a_low_rank_dim=32, v_low_rank_dim=16, gate_low_rank_dim=32)
    return RWKV7CosyLM(cfg)


def main(write):
    configure_optimizer, update_learning_rate = ref_functions()
    # deepspeed.ops.adam.FusedAdam is imported inside configure_optimizer
    adam = types.ModuleType("deepspeed.ops.adam")
    adam.FusedAdam = _Recorder
    adam.DeepSpeedCPUAdam = _Recorder
    for name, mod in (("deepspeed", types.ModuleType("deepspeed")), ("deepspeed.ops", types.ModuleType("deepspeed.ops")),
                      ("deepspeed.ops.adam", adam)):
        sys.modules[name] = mod
    model = small_cosy_model()
    from rwkvtts_amd import trainer
    out = {}
    for wd in (0.0, 0.1):
        args = types.SimpleNamespace(weight_decay=wd, ds_optimizer_offload=False, learning_rate=1e-4)
        opt = configure_optimizer(model, args)
        by_id = {}
        for g in opt.param_groups:
            for p in g["params"]:
                by_id[id(p)] = (g["name"], g["my_lr_scale"], g["weight_decay"])
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        ref = [by_id[id(p)] for n, p in model.named_parameters() if p.requires_grad]
        ours = trainer.reference_param_groups(model, wd)
        bad = [(n, r, o) for n, r, o in zip(names, ref, ours) if (r[0], float(r[1]), float(r[2])) != o]
        print(f"weight_decay={wd}: {len(names)} parameters, groups {sorted(set(r[0] for r in ref))}, mismatches: {len(bad)}")
        assert not bad, bad[:5]
        tag = "wd0" if wd == 0 else "wd"
        out[f"names_{tag}"] = np.array(names)
        out[f"group_{tag}"] = np.array([r[0] for r in ref])
        out[f"scale_{tag}"] = np.array([r[1] for r in ref], dtype=np.float64)
        out[f"decay_{tag}"] = np.array([r[2] for r in ref], dtype=np.float64)
    if write:
        np.savez(os.path.join(GOLD, "optimizer_groups.npz"), **out)

    total, warm, lr, lr_final = 1000, 100, 1e-4, 1e-5
    steps = [0, 1, 37, 99, 100, 101, 250, 550, 999, 1000, 1500]
    fake = types.SimpleNamespace(param_groups=[{"name": "lr_1x", "my_lr_scale": 1.0, "weight_decay": 0.0, "params": [], "lr": 0},
                                               {"name": "lr_2x", "my_lr_scale": 2.0, "weight_decay": 0.0, "params": [], "lr": 0},
                                               {"name": "lr_decay", "my_lr_scale": 1.0, "weight_decay": 0.1, "params": [], "lr": 0}])
    args = types.SimpleNamespace(weight_decay=0.1)
    rec = {g["name"]: [] for g in fake.param_groups}
    for s in steps:
        update_learning_rate(fake, s, total, warm, lr, lr_final, args, False)
        for g in fake.param_groups:
            rec[g["name"]].append(g["lr"])
        base = trainer.cosine_warmup_decay(s, total, warm, lr, lr_final)
        assert abs(base - fake.param_groups[0]["lr"]) < 1e-15 and abs(2 * base - fake.param_groups[1]["lr"]) < 1e-15, (s, base)
    print("cosine schedule agrees at steps", steps)
    if write:
        np.savez(os.path.join(GOLD, "lr_schedule.npz"), steps=np.array(steps), total_steps=total, warmup_steps=warm, lr=lr,
                 lr_final=lr_final, **{k: np.array(v, dtype=np.float64) for k, v in rec.items()})
        print("written")


if __name__ == "__main__":
    main("--write" in sys.argv)
