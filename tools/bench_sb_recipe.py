"""The SpeechBrain recipe's train step on one MI355X (secondary measurement, not the headline line).

Model: wav2vec2 large (LeBenchmark 7K-large shape: hidden 1024, 24 layers, 16 heads, FFN 4096, layer-norm feature encoder,
stable layer norm) + the yaml's head (3 x [Linear 1024 -> BatchNorm1d -> LeakyReLU -> Dropout 0.15] -> Linear 76), batch of the
yaml (batch_size 32) of 10 s utterances.  Two modes: freeze_wav2vec True (the yaml's default: encoder forward only, Adadelta on
the head) and False (full backward, Adam on wav2vec2).  Prints utterances/s per mode and the per-kernel time of the head's
row-wise kernels with their HBM roofline fraction.

usage: python tools/bench_sb_recipe.py [B=32] [steps=10]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from ssak_amd import hip
from ssak_amd.config import Wav2Vec2Config
from ssak_amd.model import Wav2Vec2ForCTC
from ssak_amd.sb_head import Brain, CTCHead

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T = 160000
cfg = Wav2Vec2Config(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                     feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True, mask_time_prob=0.0)
g = torch.Generator().manual_seed(69)


def make_model():
    model = Wav2Vec2ForCTC(cfg, freeze_feature_encoder=True, seed=69)
    sd = {}
    for name, (off, n, shape) in model.layout.items():
        if name.endswith("layer_norm.weight"):
            t = torch.ones(shape)
        elif name.endswith(".bias"):
            t = torch.zeros(shape)
        elif name.endswith("masked_spec_embed"):
            t = torch.rand(shape, generator=g)
        elif ".conv.weight" in name or name.endswith("original1"):
            t = torch.randn(shape, generator=g) * (2.0 / (shape[1] * shape[2])) ** 0.5
        else:
            t = torch.randn(shape, generator=g) * 0.02
        sd[name] = t
    v = sd["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    sd["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    model.load_state_dict(sd)
    return model


wavs = (torch.randn(B, T, generator=g) * 0.1).cuda()
wav_lens = torch.ones(B)
tokens = torch.randint(1, 76, (B, 120), generator=g)
tok_lens = torch.ones(B)
out = {}
model = make_model()
for freeze in (True, False):
    head = CTCHead(1024, 1024, 76, seed=1)
    brain = Brain(model, head, freeze_wav2vec=freeze)
    for _ in range(3):
        loss = brain.fit_batch(wavs, wav_lens, tokens, tok_lens, check_finite=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = brain.fit_batch(wavs, wav_lens, tokens, tok_lens, check_finite=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out["frozen" if freeze else "unfrozen"] = {"ms_per_step": round(dt * 1e3, 2), "utt_per_s": round(B / dt, 1),
                                                "loss": round(float(loss), 4)}
    model.set_grad_ready_callback(None)

# the head alone, kernel by kernel (HIP events on the launch stream)
head = CTCHead(1024, 1024, 76, seed=1)
F = model.num_frames(T)
M = B * F
feats = torch.randn(B, F, 1024, generator=g).to(torch.bfloat16).cuda()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


import ctypes as C
a = torch.randn(M, 1024, generator=g).to(torch.bfloat16).cuda()
y, dx = torch.empty_like(a), torch.empty_like(a)
mean, rstd = torch.empty(1024, device="cuda"), torch.empty(1024, device="cuda")
gam, bet = torch.ones(1024, device="cuda"), torch.zeros(1024, device="cuda")
dgm, dbt = torch.empty(1024, device="cuda"), torch.empty(1024, device="cuda")
rm, rv = torch.zeros(1024, device="cuda"), torch.ones(1024, device="cuda")
ws = torch.empty(hip.lib.ssak_batchnorm_workspace_bytes(1024), dtype=torch.uint8, device="cuda")
bytes_el = M * 1024 * 2
t_fwd = timed(lambda: hip.check(hip.lib.ssak_batchnorm_act_fwd(hip.ptr(a), hip.ptr(y), M, 1024, hip.ptr(gam), hip.ptr(bet), hip.ptr(rm),
                                                              hip.ptr(rv), 0.1, 1e-5, 1, 0.01, 0.15, C.c_uint64(5), 0, hip.ptr(mean),
                                                              hip.ptr(rstd), None, hip.ptr(ws), ws.numel(), hip.stream())))
t_bwd = timed(lambda: hip.check(hip.lib.ssak_batchnorm_act_bwd(hip.ptr(y), hip.ptr(a), hip.ptr(dx), M, 1024, hip.ptr(gam), hip.ptr(bet),
                                                              hip.ptr(mean), hip.ptr(rstd), 0.01, 0.15, C.c_uint64(5), 0, hip.ptr(dgm),
                                                              hip.ptr(dbt), None, None, hip.ptr(ws), ws.numel(), hip.stream())))
uws = torch.empty(hip.lib.ssak_utt_norm_workspace_bytes(B), dtype=torch.uint8, device="cuda")
stats = torch.empty(B, 2, device="cuda")
t_un = timed(lambda: hip.check(hip.lib.ssak_utt_norm_fwd(hip.ptr(feats), hip.ptr(y), B, F * 1024, 1, 1e-5, hip.ptr(stats), hip.ptr(uws),
                                                        uws.numel(), hip.stream())))
head.train()


def head_step():
    lg = head(feats)
    head.backward(lg, need_input_grad=False)  # any fp32 [B, F, Vp] tensor serves as dlogits for timing


t_head = timed(head_step, 10)
out["head_kernels_us"] = {
    "batchnorm_act_fwd (3 passes: 2 reads + 1 write)": [round(t_fwd, 1), round(3 * bytes_el / t_fwd * 1e-6, 2)],
    "batchnorm_act_bwd (5 passes: 4 reads + 1 write)": [round(t_bwd, 1), round(5 * bytes_el / t_bwd * 1e-6, 2)],
    "utt_norm_fwd (3 passes)": [round(t_un, 1), round(3 * bytes_el / t_un * 1e-6, 2)],
    "head forward + backward (frozen encoder)": [round(t_head, 1), None],
    "columns": "[microseconds, algorithmic TB/s]",
}
out["config"] = {"B": B, "T": T, "frames": F, "model": "wav2vec2-large (24 x 1024) + DNN head 3 x 1024 -> 76"}
print(json.dumps(out))
