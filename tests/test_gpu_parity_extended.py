"""GPU parity tests added in the last (GPU-less) session of round 1.  They sort after tests/test_gpu_parity.py on
purpose: ``pytest -x`` reaches them only after the suite that has already run green on hardware.

* the seeded index_max edge cases of tests/test_oracle_differential.py through the HIP kernel,
* the part-segmentation and autoencoder forwards at the BASELINE sizes of configs[2] / configs[3],
* size-independent properties of the classifier forward at the bench.py workload shape,
* the plain-C99 host program (tests/cabi/cabi_hotpath.c) driving the C ABI on its own HIP stream."""
import subprocess
from argparse import Namespace

import numpy as np
import pytest
import torch

import test_gpu_parity as T
from conftest import assert_close_rms
from test_gpu_parity import DEV, cu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flavour", ["plain", "ties", "special", "empty", "low"])
def test_index_max_random_edge_cases_vs_compiled_reference(flavour):
    """The seeded cases of tests/test_oracle_differential.py (ties, NaN / +-inf / <= -1000 / signed zeros, empty nodes,
    nothing above -1000; odd and tiny sizes) through the HIP kernel, against the reference's own compiled
    index_max.cpp (oracle/_ref) where it was built, else against the C restatement pinned to it."""
    from oracle import cpu_oracle as O
    from sonet_hip import ops
    from test_oracle_differential import SHAPES, _index_max_case
    for shape in SHAPES:
        B, C, N, K = shape
        data, index = _index_max_case(1000 * SHAPES.index(shape) + len(flavour), B, C, N, K, flavour)
        ref = O.ref_index_max(data, index, K) if O.ref_module() is not None else O.index_max(data, index, K)
        out = ops.index_max(cu(data), cu(index), K)
        np.testing.assert_array_equal(out.cpu().numpy(), ref, err_msg="%s %s" % (shape, flavour))


@pytest.mark.parametrize("mode", ["h3", "x3", "f32"])
def test_segmenter_forward_golden_at_the_configs2_size(mode):
    T.test_segmenter_forward_golden(mode, "segmenter_b2_n1024")


@pytest.mark.parametrize("mode", ["h3", "x3", "f32"])
def test_autoencoder_forward_and_chamfer_golden_at_the_configs3_size(mode):
    T.test_autoencoder_forward_and_chamfer_golden(mode, "autoencoder_b2_n5000")


def test_every_cloud_of_the_bench_batch_matches_the_oracle():
    """The headline workload of bench.py (64 clouds x 5000 points, its seeds, its weights): node ids of ALL 64 clouds bit-exact and the
    features / scores of ALL 64 clouds within 1e-5 of the oracle (bench.py itself checks four clouds per run; the pooled kernel is
    otherwise compared with the store kernel + index_max)."""
    import bench
    from models import networks as NW
    from oracle import cpu_oracle as O
    from sonet_hip import synth
    B, N = 64, 5000
    dev = torch.device(DEV)
    opt = bench.make_opt(dev, B, N)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    enc_sd = {k: v.clone() for k, v in synth.fill_state_dict_(enc.state_dict(), 0).items()}
    cls_sd = {k: v.clone() for k, v in synth.fill_state_dict_(cls.state_dict(), 1).items()}
    enc.to(dev).eval()
    cls.to(dev).eval()
    inp = synth.make_inputs(B, N, seed=100, device=dev)
    with torch.no_grad():
        feat = enc(inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"], is_train=False)
        score = cls(feat)
    ref = O.encoder_forward(enc_sd, inp["pc"].cpu(), inp["sn"].cpu(), inp["node"].cpu(), inp["node_knn_I"].cpu(),
                            use_ref_index_max=O.ref_module() is not None)
    np.testing.assert_array_equal(enc.min_idx.cpu().numpy(), ref["min_idx"])
    assert_close_rms(enc.first_pn_out_masked_max.cpu().numpy(), ref["first_pn_out_masked_max"].numpy(), 1e-5, "pooled first PointNet, 64 clouds")
    assert_close_rms(feat.cpu().numpy(), ref["feature"].numpy(), 1e-5, "feature, 64 clouds")
    assert_close_rms(score.cpu().numpy(), O.classifier_forward(cls_sd, ref["feature"]).numpy(), 1e-5, "score, 64 clouds")


def test_forward_properties_at_the_benchmark_shape():
    """Size-independent properties of the classifier forward at the bench.py workload shape (5000 points, 8x8 SOM, k=3),
    no oracle needed: (1) a batch shard computed alone equals the same clouds inside the full batch -- the property that
    lets bench.py / data-parallel inference shard over ranks with no data-path collective; (2) permuting
    the points of every cloud permutes the node ids and leaves every pooled feature unchanged up to the order of the
    cluster-mean sums."""
    from models import networks as NW
    from sonet_hip import synth
    B, N = 16, 5000
    opt = Namespace(gpu_id=0, device=torch.device(DEV), batch_size=B, input_pc_num=N, surface_normal=True, feature_num=1024,
                    activation="relu", normalization="batch", dropout=0.7, node_num=64, k=3, som_k=9, som_k_type="avg",
                    bn_momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=0.6, classes=40)
    enc, cls = NW.Encoder(opt), NW.Classifier(opt)
    synth.fill_state_dict_(enc.state_dict(), 21)
    synth.fill_state_dict_(cls.state_dict(), 22)
    enc.to(DEV).eval()
    cls.to(DEV).eval()
    inp = synth.make_inputs(B, N, seed=77, device=DEV)
    pc, sn, node, knn = inp["pc"], inp["sn"], inp["node"], inp["node_knn_I"]
    with torch.no_grad():
        feat = enc(pc, sn, node, knn, is_train=False).clone()
        score = cls(feat).clone()
        min_idx = enc.min_idx.clone()
        pooled = enc.first_pn_out_masked_max.clone()
        som_node = enc.som_node.clone()
        occupied = enc._lazy["a"].count > 0                                   # B x M
        # (1) shard alone == slice of the full batch
        h = B // 2
        for lo, hi in ((0, h), (h, B)):
            f_s = enc(pc[lo:hi].contiguous(), sn[lo:hi].contiguous(), node[lo:hi].contiguous(), knn[lo:hi].contiguous(), is_train=False)
            assert torch.equal(enc.min_idx, min_idx[lo:hi])
            assert_close_rms(f_s.cpu().numpy(), feat[lo:hi].cpu().numpy(), 1e-6, "feature of a shard vs the full batch")
            assert_close_rms(cls(f_s).cpu().numpy(), score[lo:hi].cpu().numpy(), 1e-6, "score of a shard vs the full batch")
        # (2) point permutation
        gen = torch.Generator().manual_seed(5)
        perm = torch.stack([torch.randperm(N, generator=gen) for _ in range(B)]).to(DEV)          # B x N
        idx3 = perm.unsqueeze(1).expand(B, 3, N)
        f_p = enc(torch.gather(pc, 2, idx3).contiguous(), torch.gather(sn, 2, idx3).contiguous(), node, knn, is_train=False)
        want = torch.gather(min_idx.view(B, 3, N), 2, idx3).reshape(B, 3 * N)                     # slot-major: [b][s*N + i] = old[b][s*N + perm[i]]
        assert torch.equal(enc.min_idx, want)
        assert torch.equal(enc._lazy["a"].count > 0, occupied)
        assert_close_rms(enc.som_node.cpu().numpy(), som_node.cpu().numpy(), 1e-6, "cluster means under a point permutation")
        # an EMPTY node takes the features of point copy 0 (the reference's gather at index 0 * mask_row_max,
        # models/networks.py:185), which a permutation legitimately changes: compare the occupied nodes only
        occ = occupied.unsqueeze(1).to(pooled.dtype)
        assert_close_rms((enc.first_pn_out_masked_max * occ).cpu().numpy(), (pooled * occ).cpu().numpy(), 1e-5,
                         "pooled features under a point permutation")
        if bool(occupied.all()):
            assert_close_rms(f_p.cpu().numpy(), feat.cpu().numpy(), 1e-5, "feature under a point permutation")



@pytest.mark.parametrize("shape", [(3, 1000, 64, 3, 40), (2, 5000, 64, 3, 384), (4, 257, 16, 2, 7), (1, 64, 100, 1, 33)])
def test_c_host_program_matches_the_oracle_on_the_gpu(tmp_path, shape):
    from test_cabi_c_host import build
    exe = build(tmp_path)
    p = subprocess.run([exe] + [str(v) for v in shape], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    assert p.stdout.strip().endswith("OK")
