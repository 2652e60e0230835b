"""-m gpu: parity at the BENCHMARK's shapes and execution mode.

The golden / oracle tests run 1-3 tiles.  bench.py runs cfg 2 at N = 32 (fp32, original, 5 types) and cfg 3 at N = 64
(bf16, fast, 6 types) through the default engine and the two-stream TilePipeline: M-tail tiles, per-sample base offsets,
the XCD mapping and the pipeline's ping-pong buffers at those sizes are exercised here.
  * every tile of the full batch is BIT-equal to its own N = 1 run (same kernels, same accumulation order: a tile's result
    cannot depend on its batch position or on the batch size);
  * tiles first / last / middle are within 1e-3 of the torch fp32 oracle (BASELINE north_star tolerance) -- the oracle is
    only run on those, it takes ~1.5 s per tile on the CPU;
  * TilePipeline.submit (network on the main stream, post-processing on the side stream, D2H to pinned memory) returns
    the same instance maps / records as the sequential path at N = 32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(mode, nt, seed, dtype="fp32", max_batch=32):
    from hover_net_amd import net_desc
    from hover_net_amd.synth import synth_state_dict

    sd = synth_state_dict(mode, nt, seed=seed)
    net = net_desc.create_model(mode=mode, nr_types=nt, input_ch=3)
    net.load_state_dict(sd, strict=True)
    net.max_batch = max_batch
    net.compute_dtype = dtype
    return sd, net.to("cuda").eval()


def test_cfg2_batch32_bit_equal_to_single_tile_runs_and_oracle():
    from hover_net_amd import run_desc
    from hover_net_amd.synth import synth_tiles
    from oracle import net_torch

    sd, net = _net("original", 5, seed=0)
    tiles = torch.from_numpy(synth_tiles(32, 270, seed=1))
    full = run_desc.infer_step_device(tiles, net).cpu().clone()
    eng = net.engine(32)
    logits = {k: v[:32].cpu().clone() for k, v in eng.logits.items()}
    assert full.shape == (32, 80, 80, 4)
    for i in range(32):
        one = run_desc.infer_step_device(tiles[i:i + 1], net).cpu()
        assert torch.equal(one[0], full[i]), "tile %d differs between N=1 and N=32" % i
    pick = [0, 15, 16, 31]
    want = net_torch.forward(sd, tiles[pick].permute(0, 3, 1, 2).float(), "original")
    for k, v in want.items():
        err = (logits[k][pick] - v).abs().max().item()
        assert err <= 1e-3, (k, err)
    wpm = net_torch.infer_epilogue(want)
    assert (full[pick][..., 1:] - wpm[..., 1:]).abs().max().item() <= 1e-3


def test_cfg3_batch64_bf16_bit_equal_to_single_tile_runs():
    from hover_net_amd import run_desc
    from hover_net_amd.synth import synth_tiles
    from oracle import net_torch

    sd, net = _net("fast", 6, seed=0, dtype="bf16", max_batch=64)
    tiles = torch.from_numpy(synth_tiles(64, 256, seed=1))
    full = run_desc.infer_step_device(tiles, net).cpu().clone()
    assert full.shape == (64, 164, 164, 4)
    for i in (0, 1, 31, 32, 62, 63):
        one = run_desc.infer_step_device(tiles[i:i + 1], net).cpu()
        assert torch.equal(one[0], full[i]), "tile %d differs between N=1 and N=64" % i
    # bf16 against the fp32 oracle on two tiles: the declared cfg-3 tolerance of tests/test_gpu_bf16.py
    pick = [0, 63]
    want = net_torch.infer_epilogue(net_torch.forward(sd, tiles[pick].permute(0, 3, 1, 2).float(), "fast"))
    assert (full[pick][..., 1] - want[..., 1]).abs().max().item() < 2e-2
    assert ((full[pick][..., 1] >= 0.5) == (want[..., 1] >= 0.5)).float().mean().item() > 0.995


def test_cfg2_fp32_fast_mode_batch64_bit_equal():
    """fp32 kernels at the cfg-3 geometry (fast mode, 164^2 outputs, N = 64)."""
    from hover_net_amd import run_desc
    from hover_net_amd.synth import synth_tiles

    _sd, net = _net("fast", 6, seed=2, max_batch=64)
    tiles = torch.from_numpy(synth_tiles(64, 256, seed=3))
    full = run_desc.infer_step_device(tiles, net).cpu().clone()
    for i in (0, 31, 32, 63):
        one = run_desc.infer_step_device(tiles[i:i + 1], net).cpu()
        assert torch.equal(one[0], full[i])


def test_pipeline_batch32_equals_sequential_with_host_output():
    from hover_net_amd import post_proc, run_desc
    from hover_net_amd.pipeline import TilePipeline
    from hover_net_amd.synth import synth_pred_maps, synth_tiles
    from oracle import postproc as O

    _sd, net = _net("original", 5, seed=0)
    structured_np = synth_pred_maps(32, 80, 80, 5, seed=100, k_lo=2, k_hi=8)[0]
    structured = torch.from_numpy(structured_np).to("cuda")
    pipe = TilePipeline(net, nr_types=5, return_centroids=True)
    batches = [torch.from_numpy(synth_tiles(32, 270, seed=1 + j)).pin_memory() for j in range(3)]
    got_net, got_ex = [], []
    for t in batches:                                         # back-to-back: slot reuse + overlap with the next network pass
        o = pipe.submit(t, to_host=True)
        pipe.side.synchronize()
        assert all((not x.is_cuda) and x.is_pinned() for x in o)
        got_net.append(tuple(x.clone() for x in o))
        o = pipe.submit(t, extra_maps=structured, to_host=True)
        pipe.side.synchronize()
        got_ex.append(tuple(x.clone() for x in o))
    # without the host syncs in between (what bench.py does), only the last result is inspected
    for t in batches:
        last = pipe.submit(t, extra_maps=structured, to_host=True)
    pipe.wait()
    want_ex = post_proc.process_batch_device(structured, 5, True)
    np.testing.assert_array_equal(want_ex[0].cpu().numpy(), O.proc_batch(structured_np))      # bit-exact vs the C oracle at N = 32
    for j, t in enumerate(batches):
        pred = run_desc.infer_step_device(t, net)
        want = post_proc.process_batch_device(pred, 5, True)
        for a, b in zip(got_net[j], want):
            assert torch.equal(a, b.cpu())
        for a, b in zip(got_ex[j], want_ex):
            assert torch.equal(a, b.cpu())
    for a, b in zip(last, want_ex):
        assert torch.equal(a, b.cpu())
