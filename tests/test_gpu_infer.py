"""GPU: the inference call conventions of ssak/infer/transformers_infer.py -- the time-chunking rule for long inputs."""
import dataclasses

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("topology", ["base", "xlsr"])
def test_compute_logits_chunks_long_inputs_like_the_reference(topology):
    """Inputs longer than ``max_duration`` samples are cut on the sample axis AFTER normalisation + padding and the chunks'
    logits concatenated on the frame axis (transformers_infer.py:259-265; chunks are independent, context is lost at the
    seams).  A small ``max_duration`` exercises the rule: the result must equal the independent forwards of the slices
    (with the sliced attention mask for the layer-norm topology), and differ from the unchunked forward."""
    import ssak_amd.hip as hip
    from oracle import w2v2_ref as R
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.infer import MAX_SAMPLES, transformers_compute_logits
    from ssak_amd.model import Wav2Vec2ForCTC
    assert MAX_SAMPLES == 2240400  # transformers_infer.py:190
    kw = {} if topology == "base" else dict(feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True)
    oc = R.W2V2Config.tiny(**kw).deterministic()
    d = dataclasses.asdict(oc)
    d.pop("initializer_range")
    model = Wav2Vec2ForCTC(Wav2Vec2Config(**d)).eval()
    model.load_state_dict(R.init_params(oc, 3))
    rng = np.random.default_rng(0)
    batch = [rng.standard_normal(30000).astype(np.float32), rng.standard_normal(17000).astype(np.float32)]
    step = 12000
    got = transformers_compute_logits(model, None, batch, max_duration=step)
    whole = transformers_compute_logits(model, None, batch)
    # by hand: normalise whole utterances, pad to the longest, slice, forward each slice on its own
    lens = torch.tensor([30000, 17000], dtype=torch.int32).cuda()
    xpad = np.zeros((2, 30000), np.float32)
    xpad[1, :17000] = batch[1]
    xpad[0] = batch[0]
    xn = hip.wave_normalize(torch.tensor(xpad).cuda(), lens)
    assert np.abs(xn.cpu().numpy() - R.zero_mean_unit_var_norm(batch)).max() < 1e-5
    parts = []
    for s in range(0, 30000, step):
        chunk = xn[:, s:s + step].contiguous()
        cl = (lens - s).clamp(min=0, max=chunk.shape[1]) if topology == "xlsr" else None
        parts.append(model(chunk, lengths=cl).logits.cpu())
    want = torch.cat(parts, dim=1)
    frames = [model.num_frames(n) for n in (12000, 12000, 6000)]
    assert got.shape == want.shape == (2, sum(frames), oc.vocab_size) and got.dtype == torch.float32 and not got.is_cuda
    assert torch.allclose(got, want, atol=1e-5, equal_nan=True)  # (frames behind a fully padded chunk are undefined)
    assert whole.shape[1] == model.num_frames(30000) != got.shape[1]  # one frame is lost at each seam
    assert float((whole[0, :frames[0] - 8] - got[0, :frames[0] - 8]).abs().max()) > 1e-3  # context differs -> different logits
