import sys, time, torch
sys.path.insert(0, ".")
from ssak_amd.config import Wav2Vec2Config
from ssak_amd.model import Wav2Vec2ForCTC
m = Wav2Vec2ForCTC(Wav2Vec2Config())
g = torch.Generator().manual_seed(0)
sd = {n: (torch.randn(shape, generator=g) * 0.02) for n, (off, k, shape) in m.layout.items()}
v = sd["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
sd["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
m.load_state_dict(sd)
for _ in range(5): m.sync_weights(full=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): m.sync_weights(full=False)
e1.record(); torch.cuda.synchronize()
print("posconv prepare (colnorm + finalize + materialize): %.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
