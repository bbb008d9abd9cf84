"""The generation pipeline (difffacto_amd/pipeline.py): front end (draws + dfx_sample_latents + dfx_shape_ctx_prepare) captured in a
hipGraph and replayed on a side stream beside the previous batch's chain.  Same bits as the plain call sequence with the same draws,
whatever the mode; consecutive batches differ (graph-safe Philox offsets of the draws, per-batch chain seed)."""
import numpy as np
import pytest
import torch

from difffacto_amd import synth

pytestmark = pytest.mark.gpu


def _parts(T, prec="bf16"):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from difffacto_amd.engine import DenoiserEngine
    from difffacto_amd.latents import LatentSampler
    W = {k: torch.from_numpy(v) for k, v in synth.make_denoiser_weights(seed=0).items()}
    eng = DenoiserEngine(W, num_timesteps=T, precision=prec)
    sampler = LatentSampler(synth.make_latent_weights(seed=0), noise_scale=100.0)
    return eng, sampler


@pytest.mark.parametrize("B,N", [(1, 2048), (6, 2048), (3, 8192)])
def test_pipeline_modes_agree_with_the_plain_call_sequence(B, N):
    from difffacto_amd.pipeline import SamplingPipeline
    T, nb = 12, 4
    eng, sampler = _parts(T)
    valid = torch.from_numpy(synth.make_latents(B, seed=B)[3].copy()).cuda()
    # reference: the plain sequence, draws from the default generator in the pipeline's order
    torch.cuda.manual_seed(77)
    ref = []
    for i in range(nb):
        w = torch.empty(B, 256, 4, device="cuda").normal_()
        an = torch.empty(B, 32, device="cuda").normal_()
        lat = sampler.sample_latents(w, an, valid, K=1, npoints=N)
        ctx = eng.prepare_shapes(lat["part_code"], lat["params"][:, :3], lat["params"][:, 3:], lat["valid_id"])
        ref.append(eng.sample_chain(ctx, lat["seg_mask"], seed=5 + i, shape_offset=3)[0].clone())
    assert not torch.equal(ref[0], ref[1]) and all(torch.isfinite(r).all() for r in ref)
    for use_graph, overlap in ((False, False), (True, False), (False, True), (True, True)):
        pipe = SamplingPipeline(eng, sampler, B, N, valid, use_graph=use_graph, overlap=overlap)   # (graph mode warms up with draws of its own)
        torch.cuda.synchronize()
        torch.cuda.manual_seed(77)
        got = [p.clone() for p in pipe.run(nb, seed0=5, shape_offset=3, time_chain=True)]
        torch.cuda.synchronize()
        assert len(got) == nb and len(pipe.last_chain_events) == nb
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a, b), (use_graph, overlap, i, (a - b).abs().max().item())
        # a second run continues the generator's stream: new clouds
        again = [p.clone() for p in pipe.run(2, seed0=5, shape_offset=3)]
        assert not torch.equal(again[0], got[0])
    eng.close()
