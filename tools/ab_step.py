"""A/B of optional paths inside ONE process on ONE box (boxes of the pool differ by ~5 %, more than most switches are worth):
the 0.4B Spark training step with each switch off and on, interleaved, median of the step times."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import backbone, fused, losses, trainer
from rwkvtts_amd.layouts import synthetic_spark_batch
from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
dev = torch.device("cuda:0")
base = backbone.config_0p4b()
kw = {k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}
model = RWKV7ForSpeech(RWKV7SpeechConfig(**kw)).init_weights(seed=0).to(device=dev, dtype=torch.bfloat16).train()
tr = trainer.DataParallelTrainer(model, lr=1e-4, warmup_steps=10, total_steps=1000)
it = [0]


def steps(n):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.step(**synthetic_spark_batch(model, 8, 4096, seed=1234 + it[0]))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        it[0] += 1
    return ts


switches = [("head: logits in a buffer padded to 256 columns, logits GEMM on the own kernel (losses.PADDED_HEAD)", losses, "PADDED_HEAD", False, True),
            ("weight gradient of [W_a ; W_b] (N = 576) on the own kernel rwkv7_wgrad_mid_bf16", fused, "MID_WGRAD", False, True),
            ("low-rank down projections: lerp as the A prologue of the own MFMA kernel (csrc/lora_down.hip)", fused, "LORA_DOWN_DIRECT", False, True),
            ("channel-mix backward: W_value^T through the own transpose kernel", fused, "TRANSPOSE_KERNEL", False, True),
            ("time-mix side: add + LayerNorm + three lerps one-pass forward (with mix_lora)", fused, "FUSED_ADD_LN_MIX_LORA_FWD", False, True),
            ("parameter-gradient partials: column sums as one launch (sum_slabs tall shape)", fused, "COLSUM_KERNEL", False, True),
            ("round 4 knobs: one-pass backward partials 2048 -> 4096 workgroups", fused, "_ADD_LN_MIX_BWD_BLOCKS", 2048, 4096),
            ("round 4 knobs: one-pass backward partials 2048 -> 8192 workgroups", fused, "_ADD_LN_MIX_BWD_BLOCKS", 2048, 8192),
            ("round 4 knobs: one-pass stages run of 4 -> 8 rows", fused, "_ADD_LN_MIX_RUN", 4, 8),
            ("round 4 knobs: one-pass stages run of 4 -> 2 rows", fused, "_ADD_LN_MIX_RUN", 4, 2),
            ("round 4 knobs: parameter-gradient partials 1024 -> 2048 workgroups", fused, "_BWD_BLOCKS", 1024, 2048),
            ("round 4 knobs: parameter-gradient partials 1024 -> 4096 workgroups", fused, "_BWD_BLOCKS", 1024, 4096),
            ("round 4 knobs: mix backward 1024 -> 2048 workgroups", fused, "_MIX_BWD_BLOCKS", 1024, 2048),
            ("round 4 knobs: mix backward 1024 -> 4096 workgroups", fused, "_MIX_BWD_BLOCKS", 1024, 4096),
            ("low-rank branches: down projections through the lerp (one GEMM on the LayerNorm output)", fused, "FUSED_MIX_LORA", False, True),
            ("output projection with the residual add as its epilogue (own kernel)", fused, "FUSED_OPROJ_ADD", False, True),
            ("channel mix: activation inside both GEMMs (own kernel, generation 4)", fused, "FUSED_CMIX", False, True),
            ("relu^2 backward in the value dgrad GEMM (own kernel)", fused, "FUSED_RELUSQ_VALUE_BWD", False, True),
            ("add+LN+six lerps: one-pass forward (backward as two kernels)", backbone, "FUSED_ADD_LN_MIX6_FWD", False, True),
            ("tmix_post backward -> prepare backward: compact hand-off (dt + per-head scalars)", fused, "COMPACT_POST_BWD", False, True),
            ("side-stream weight gradients drained before the WKV7 backward", fused, "WGRAD_SYNC_BEFORE_SCAN", False, True),
            ("weight gradients on a side stream", fused, "WGRAD_SIDE_STREAM", False, True),
            ("channel-mix key GEMM with relu^2 epilogue (own kernel)", fused, "FUSED_KEY_RELUSQ", False, True),   # measured: -0.13 ms, off
            ("add+LN+mix one pass (channel-mix side)", backbone, "FUSED_ADD_LN_MIX1", False, True),
            ("low-rank weight gradients: skinny kernel", fused, "SKINNY_WGRAD", False, True),
            ("value projection + value-residual branch as one node", backbone, "DUAL_LINEAR_XV", False, True),
            ("v_first gradient summed layer by layer in the prepare backward", fused, "CHAIN_VFIRST_GRAD", False, True),
            ("parameter-gradient partials 2048 -> 1024 workgroups", fused, "_BWD_BLOCKS", 2048, 1024),
            ("parameter-gradient partials 1024 -> 512 workgroups", fused, "_BWD_BLOCKS", 1024, 512),
            ("one-pass backward partials 2048 -> 1024 workgroups", fused, "_ADD_LN_MIX_BWD_BLOCKS", 2048, 1024),
            ("mix backward runs 4 -> 8 rows", fused, "_MIX_BWD_ROWS", 4, 8),
            ("forward stages 32768 -> 16384 workgroups", fused, "_FWD_BLOCKS", 32768, 16384),
            ("forward stages 32768 -> 8192 workgroups", fused, "_FWD_BLOCKS", 32768, 8192),
            ("mix forward 2048 -> 4096 workgroups", fused, "_MIX_FWD_BLOCKS", 2048, 4096),
            ("mix forward 2048 -> 1024 workgroups", fused, "_MIX_FWD_BLOCKS", 2048, 1024),
            ("weight gradients 1024x1024: 8 -> 16 slabs", fused, "WGRAD_SLABS_SMALL", 8, 16),
            ("weight gradients 1024x1024: 8 -> 4 slabs", fused, "WGRAD_SLABS_SMALL", 8, 4),
            ("weight gradients 4096x1024: 4 -> 8 slabs", fused, "WGRAD_SLABS_BIG", 4, 8),
            ("weight gradients 4096x1024: 4 -> 2 slabs", fused, "WGRAD_SLABS_BIG", 4, 2),
            ("mix backward 1024 -> 2048 workgroups", fused, "_MIX_BWD_BLOCKS", 1024, 2048)]
if len(sys.argv) > 1:
    switches = [s_ for s_ in switches if any(k in s_[0] for k in sys.argv[1:])]
steps(3)
for name, mod, attr, off, on in switches:
    default = getattr(mod, attr)
    res = {off: [], on: []}
    for rep in range(int(os.environ.get("AB_REPS", "3"))):
        for v in (off, on):
            setattr(mod, attr, v)
            steps(1)
            res[v] += steps(4)
    setattr(mod, attr, default)
    med = {v: sorted(res[v])[len(res[v]) // 2] for v in res}
    print(f"{name:62s} {off!s:>5}: {med[off]:7.2f} ms   {on!s:>5}: {med[on]:7.2f} ms   ({med[on] - med[off]:+.2f} ms; default {default})", flush=True)
