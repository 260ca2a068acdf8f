"""oracle/make_golden.py -- TEST INFRASTRUCTURE: generate tests/golden/*.npz from the LIVE reference.

Run in the build container (needs /root/reference):   python oracle/make_golden.py
Each fixture holds seeded inputs and the outputs of the unmodified reference code on them
(imported through oracle/ref_harness.py).  Weights are NOT stored: both sides regenerate them
with sonet_hip.synth.fill_state_dict_(seed).  The fixtures are the pin for oracle/sonet_oracle.c
(tests/test_oracle_golden.py) and, on the GPU box where /root/reference does not exist, for the
HIP path (tests/test_*_gpu.py).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.environ.get("SONET_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # (tests regenerate into a scratch dir)
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != HERE]
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402


def _load_synth():
    spec = importlib.util.spec_from_file_location(
        "sonet_synth", os.path.join(ROOT, "so-net_amd", "sonet_hip", "synth.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = _load_synth()


def save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(path, **out)
    print("%-40s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------ index_max
def golden_index_max(ref):
    fwd, fwd_mt = ref.index_max.forward_cpu, ref.index_max.forward_multi_thread_cpu
    g = torch.Generator().manual_seed(11)
    # (1) plain random
    B, C, N, K = 3, 24, 700, 64
    data = torch.randn(B, C, N, generator=g)
    index = torch.randint(0, K, (B, N), generator=g, dtype=torch.int32)
    out = fwd(data, index, K)
    assert torch.equal(out, fwd_mt(data, index, K, 4))
    save("index_max_random", data=data, index=index, K=K, out=out)
    # (2) adversarial: ties, NaN, -inf/+inf, values <= -1000, -0.0/+0.0, empty nodes, quantised values
    B, C, N, K = 2, 16, 513, 64
    data = torch.randn(B, C, N, generator=g)
    data = (data * 4).round() / 4                                   # heavy exact ties
    data[0, 0, :] = 0.0                                             # all equal -> smallest n per node
    data[0, 1, :] = -1000.0                                         # never beats the init value
    data[0, 2, :] = -2000.0
    data[0, 3, :] = float("nan")
    data[0, 4, ::2] = float("nan")
    data[0, 5, :] = -0.0
    data[0, 5, 1::2] = 0.0                                          # +0.0 after -0.0: no update
    data[0, 6, :] = 0.0
    data[0, 6, 1::2] = -0.0
    data[0, 7, 7] = float("inf")
    data[0, 8, :] = float("-inf")
    data[0, 9, :] = -1000.0
    data[0, 9, 100] = -999.99994
    data[1, 0, :] = torch.arange(N, dtype=torch.float32)            # strictly increasing
    data[1, 1, :] = -torch.arange(N, dtype=torch.float32)           # strictly decreasing
    data[1, 2, :] = 1e-42                                           # subnormal
    data[1, 3, :] = -1e-42
    index = torch.randint(0, K - 8, (B, N), generator=g, dtype=torch.int32)   # nodes K-8.. stay empty
    index[1, :] = 5                                                 # one node owns every point
    out = fwd(data, index, K)
    assert torch.equal(out, fwd_mt(data, index, K, 3))
    save("index_max_adversarial", data=data, index=index, K=K, out=out)
    # (3) odd sizes: N not a multiple of anything, K != 64, C == 1, single point
    for tag, (B, C, N, K) in {"odd": (2, 5, 1021, 37), "tiny": (1, 1, 1, 4), "k256": (1, 3, 999, 256),
                              "k1": (2, 2, 130, 1)}.items():
        data = torch.randn(B, C, N, generator=g)
        index = torch.randint(0, K, (B, N), generator=g, dtype=torch.int32)
        save("index_max_" + tag, data=data, index=index, K=K, out=fwd(data, index, K))


# ------------------------------------------------------------------------------------------ query_topk
def golden_query_topk(ref):
    for tag, (B, N, rows, seed, kind) in {"a": (2, 257, 8, 21, "uniform"), "b": (3, 1024, 8, 22, "som"),
                                          "c": (1, 100, 4, 23, "uniform")}.items():
        M = rows * rows
        inp = synth.make_inputs(B, N, M=M, som_k=min(9, M), seed=seed, node_kind=kind)
        bs = ref.som.BatchSOM(rows, rows, 3, 0, B)
        arrays = dict(x=inp["pc"], node=inp["node"])
        for k in (1, 2, 3):
            bs.node.resize_(inp["node"].size()).copy_(inp["node"])
            mask_u, rmax_u, idx_u = bs.query_topk(inp["pc"], k)                  # reference, as is
            with ref_harness.sorted_topk():
                mask_s, rmax_s, idx_s = bs.query_topk(inp["pc"], k)              # canonical slot order
            assert torch.equal(rmax_u, rmax_s)
            assert torch.equal(mask_u.sum(1), mask_s.sum(1))
            arrays["min_idx_unsorted_k%d" % k] = idx_u
            arrays["min_idx_sorted_k%d" % k] = idx_s
            arrays["mask_row_max_k%d" % k] = rmax_s
            arrays["mask_row_sum_k%d" % k] = mask_s.sum(1)
            arrays["mask_dtype_k%d" % k] = str(mask_s.dtype)
        save("query_topk_" + tag, **arrays)


# ------------------------------------------------------------------------------------------ model forward
def golden_classifier(ref, tag, B, N, seed, node_kind, k=3, som_k=9, som_k_type="avg"):
    opt = ref_harness.make_opt(batch_size=B, input_pc_num=N, k=k, som_k=som_k, som_k_type=som_k_type,
                               classes=40)
    model = ref.classifier.Model(opt)
    synth.fill_state_dict_(model.encoder.state_dict(), seed=seed)       # tensors alias the parameters
    synth.fill_state_dict_(model.classifier.state_dict(), seed=seed + 1)
    inp = synth.make_inputs(B, N, M=opt.node_num, som_k=som_k, seed=seed, node_kind=node_kind)
    model.set_input(inp["pc"], inp["sn"], inp["label"], inp["node"], inp["node_knn_I"])
    with ref_harness.sorted_topk(), torch.no_grad():
        model.test_model()
    enc = model.encoder
    min_idx = enc.mask.argmax(dim=2)                                     # segmenter.py:90
    first = enc.first_pn_out
    save("classifier_" + tag,
         B=B, N=N, k=k, som_k=som_k, som_k_type=som_k_type, seed=seed, node_kind=node_kind,
         pc=inp["pc"], sn=inp["sn"], node=inp["node"], node_knn_I=inp["node_knn_I"], label=inp["label"],
         min_idx=min_idx, mask_row_sum=enc.mask.sum(1), som_node=enc.som_node, centers=enc.centers[:, :, ::7],
         x_decentered=enc.x_decentered[:, :, ::7],
         first_pn_out_sub=first[:, ::16, ::5], first_pn_out_rms=first.pow(2).mean().sqrt(),
         first_pn_out_masked_max=enc.first_pn_out_masked_max,
         knn_center_1=enc.knn_center_1, knn_feature_1=enc.knn_feature_1[:, ::4],
         final_pn_out=enc.final_pn_out[:, ::4], feature=enc.feature, score=model.score, loss=model.loss)
    return model, inp


def golden_layers(ref):
    """EquivariantLayer / PointResNet / KNNModule in isolation, eval and train mode."""
    g = torch.Generator().manual_seed(31)
    L = ref.layers
    arrays = {}
    # eval + train single layer 6 -> 64 with BN + relu
    x = torch.randn(2, 6, 300, generator=g)
    layer = L.EquivariantLayer(6, 64, "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(layer.state_dict(), seed=5)
    layer.eval()
    with torch.no_grad():
        arrays["eq_eval_y"] = layer(x)
    layer.train()
    y = layer(x, epoch=None)
    arrays["eq_x"] = x
    arrays["eq_train_y"] = y
    arrays["eq_train_running_mean"] = layer.norm.running_mean.clone()
    arrays["eq_train_running_var"] = layer.norm.running_var.clone()
    arrays["eq_train_num_batches_tracked"] = layer.norm.num_batches_tracked.clone()
    gy = torch.randn(y.shape, generator=g)
    xg = x.clone().requires_grad_(True)
    synth.fill_state_dict_(layer.state_dict(), seed=5)
    y2 = layer(xg)
    y2.backward(gy)
    arrays["eq_gy"] = gy
    arrays["eq_train_gx"] = xg.grad
    arrays["eq_train_gw"] = layer.conv.weight.grad
    arrays["eq_train_gb"] = layer.conv.bias.grad
    arrays["eq_train_ggamma"] = layer.norm.weight.grad
    arrays["eq_train_gbeta"] = layer.norm.bias.grad
    # momentum decay rule (layers.py:60-65)
    layer2 = L.EquivariantLayer(6, 64, "relu", "batch", 0.5, 2, 0.6)
    synth.fill_state_dict_(layer2.state_dict(), seed=6)
    layer2.train()
    layer2(x, epoch=5)
    arrays["eq_decay_momentum"] = layer2.norm.momentum
    arrays["eq_decay_running_mean"] = layer2.norm.running_mean.clone()
    # PointResNet 6 -> [64,128,256,384] eval
    pr = L.PointResNet(6, [64, 128, 256, 384], "relu", "batch", 0.1, None, 1)
    synth.fill_state_dict_(pr.state_dict(), seed=7)
    pr.eval()
    with torch.no_grad():
        arrays["prn_eval_y"] = pr(x)[:, ::8]
    save("layers", **arrays)


def golden_state_dict_keys(ref):
    """Key names and shapes of the reference modules: the checkpoint compatibility contract."""
    import json
    opt = ref_harness.make_opt(batch_size=2, input_pc_num=256)
    out = {}
    for name, mod in (("encoder", ref.networks.Encoder(opt)), ("classifier", ref.networks.Classifier(opt))):
        out[name] = {k: list(v.shape) for k, v in mod.state_dict().items()}
    L = ref.layers
    out["knnmodule"] = {k: list(v.shape) for k, v in L.KNNModule(387, (512, 512), "relu", "batch").state_dict().items()}
    out["myconv2d"] = {k: list(v.shape) for k, v in L.MyConv2d(4, 8, 1, activation="relu", normalization="batch").state_dict().items()}
    with open(os.path.join(GOLD, "state_dict_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("state_dict_keys.json  encoder %d keys, classifier %d keys" % (len(out["encoder"]), len(out["classifier"])))


GRAD_KEYS = ["first_pointnet.layers.0.conv.weight", "first_pointnet.layers.0.norm.weight", "first_pointnet.layers.2.conv.weight",
             "first_pointnet.layers.3.conv.weight", "first_pointnet.layers.3.conv.bias", "knnlayer.layers.0.conv.weight",
             "knnlayer.layers.1.norm.bias", "final_pointnet.layers.0.conv.weight", "final_pointnet.layers.1.conv.bias"]


def golden_train_step(ref, tag, B, N, seed):
    """One classifier training step of the reference (models/classifier.py:78-99), dropout off so it is
    deterministic: loss, selected gradients, BN running statistics and parameters after the Adam steps.

    Gradients are stored twice: from the reference as it is (float32 on CPU) and from the SAME reference code run
    in float64 (``grad64/``).  The float32 CPU backward of train-mode BatchNorm under the very sparse gradients of
    the arg-max pools loses ~3 digits to cancellation (its deviation from the float64 run is stored as
    ``ref32_dev/``), so the float64 run is the ground truth the HIP path is held to.

    ``route64/pool{1,2,3}`` (``route32/`` for the float32 run): the arg-max positions the run took at its three pools -- pool1 = what
    ``index_max.forward_cuda`` returned times ``mask_row_max`` (the gather index of models/networks.py:185), pool2 / pool3 = the indices
    of the two ``torch.max(..., dim=...)`` calls (models/layers.py:365, models/networks.py:197) -- recorded while the unmodified reference
    runs.  A float64 restatement with these positions FORCED must reproduce ``grad64/`` (tests/f64_classifier.py)."""
    def run(dtype):
        log = {}
        orig_max = torch.max

        def recording_max(*a, **k):
            r = orig_max(*a, **k)
            if isinstance(r, tuple) and a and torch.is_tensor(a[0]) and a[0].is_floating_point() and a[0].dim() in (3, 4):
                log["pool2" if a[0].dim() == 4 else "pool3"] = r[1].clone()
            return r
        opt = ref_harness.make_opt(batch_size=B, input_pc_num=N, dropout=0.0, classes=40)
        model = ref.classifier.Model(opt)
        synth.fill_state_dict_(model.encoder.state_dict(), seed=seed)
        synth.fill_state_dict_(model.classifier.state_dict(), seed=seed + 1)
        inp = synth.make_inputs(B, N, M=opt.node_num, som_k=opt.som_k, seed=seed, node_kind="som")
        if dtype == torch.float64:
            model.encoder.double(); model.classifier.double()
            model.input_pc, model.input_sn, model.input_node = model.input_pc.double(), model.input_sn.double(), model.input_node.double()
            model.encoder.som_builder.node = model.encoder.som_builder.node.double()
            model.optimizer_encoder = torch.optim.Adam(model.encoder.parameters(), lr=0.001)
            model.optimizer_classifier = torch.optim.Adam(model.classifier.parameters(), lr=0.001)
        # the reference's index_max extension reads float32 only (index_max.cpp:92): cast at the shim (a no-op in the float32 run),
        # indices only; the shim also records what the extension returned
        ext = ref.index_max
        fwd = ext.forward_cpu
        ref.networks.index_max = type(ext)("index_max")

        def forward_cuda(d, i, K):
            out = fwd(d.float().contiguous(), i, K)
            log["pool1_raw"] = out.clone()
            return out
        ref.networks.index_max.forward_cuda = forward_cuda
        model.set_input(inp["pc"].to(dtype), inp["sn"].to(dtype), inp["label"], inp["node"].to(dtype), inp["node_knn_I"])
        torch.max = recording_max
        try:
            with ref_harness.sorted_topk():
                model.optimize(epoch=0)
        finally:
            torch.max = orig_max
            ref.networks.index_max = ref.index_max
        row_max = orig_max(model.encoder.mask, dim=1)[0]                            # util/som.py:267
        log["pool1"] = log.pop("pool1_raw").long() * row_max.unsqueeze(1).long()    # models/networks.py:185
        return model, inp, log

    model, inp, route32 = run(torch.float32)
    model64, _, route64 = run(torch.float64)
    enc = dict(model.encoder.named_parameters())
    enc64 = dict(model64.encoder.named_parameters())
    arrays = dict(B=B, N=N, seed=seed, pc=inp["pc"], sn=inp["sn"], node=inp["node"], node_knn_I=inp["node_knn_I"],
                  label=inp["label"], loss=model.loss.detach(), loss64=model64.loss.detach(), feature=model.feature.detach(),
                  score=model.score.detach())

    for name, route in (("route32", route32), ("route64", route64)):
        assert int(route["pool1"].max()) < 32768 and int(route["pool2"].max()) < 128 and int(route["pool3"].max()) < 128
        arrays[name + "/pool1"] = route["pool1"].to(torch.int16)
        arrays[name + "/pool2"] = route["pool2"].to(torch.int8)
        arrays[name + "/pool3"] = route["pool3"].to(torch.int8)
        print("   %s: pool1 %s pool2 %s pool3 %s" % (name, tuple(route["pool1"].shape), tuple(route["pool2"].shape), tuple(route["pool3"].shape)))

    def sub(t):                                   # strided subsample keeps the fixture small (<= ~16k values per tensor)
        f = t.detach().flatten()
        return f[::max(1, f.numel() // 16384)]
    for k in GRAD_KEYS:
        arrays["grad/" + k] = sub(enc[k].grad)
        arrays["grad64/" + k] = sub(enc64[k].grad)
        arrays["after/" + k] = sub(enc[k])
        a, r = enc[k].grad.double(), enc64[k].grad
        arrays["ref32_dev/" + k] = ((a - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item()
        print("   %-45s float32 reference vs its own float64 run: rel-rms %.2e" % (k, arrays["ref32_dev/" + k]))
    cls = dict(model.classifier.named_parameters())
    arrays["grad/cls.fc1.linear.weight"] = sub(cls["fc1.linear.weight"].grad)
    arrays["grad64/cls.fc1.linear.weight"] = sub(dict(model64.classifier.named_parameters())["fc1.linear.weight"].grad)
    arrays["dead_grad_count"] = sum(1 for p in model.encoder.parameters() if p.grad is None)
    sd = model.encoder.state_dict()
    for k in ("first_pointnet.layers.1.norm.running_mean", "first_pointnet.layers.1.norm.running_var",
              "knnlayer.layers.0.norm.running_var", "final_pointnet.layers.0.norm.running_mean"):
        arrays["bn/" + k] = sd[k]
    save("train_step_" + tag, **arrays)


def golden_segmenter(ref, tag, B, N, seed):
    """Reference part-segmentation forward (models/segmenter.py:79-109 + networks.Segmenter), eval mode."""
    opt = ref_harness.make_opt(batch_size=B, input_pc_num=N, classes=50, dropout=0.6, som_k_type="center")
    model = ref.segmenter.Model(opt)
    synth.fill_state_dict_(model.encoder.state_dict(), seed=seed)
    synth.fill_state_dict_(model.segmenter.state_dict(), seed=seed + 1)
    inp = synth.make_inputs(B, N, M=opt.node_num, som_k=opt.som_k, seed=seed, node_kind="som")
    g = torch.Generator().manual_seed(seed)
    label = torch.randint(0, 16, (B,), generator=g)
    seg = torch.randint(0, 50, (B, N), generator=g)
    model.set_input(inp["pc"], inp["sn"], label, seg, inp["node"], inp["node_knn_I"])
    model.encoder.eval(); model.segmenter.eval()
    with ref_harness.sorted_topk(), torch.no_grad():
        model.forward(is_train=False)
    enc = model.encoder
    min_idx = enc.mask.argmax(dim=2)
    idx384 = min_idx.unsqueeze(1).expand(B, 384, min_idx.shape[1])
    bb = torch.gather(enc.first_pn_out_masked_max, 2, idx384)
    save("segmenter_" + tag, B=B, N=N, seed=seed, pc=inp["pc"], sn=inp["sn"], node=inp["node"], node_knn_I=inp["node_knn_I"],
         label=label, min_idx=min_idx, feature_max_first_pn_out=bb[:, ::8], score_segmenter=model.score_segmenter,
         segmenter_keys=np.array(sorted(model.segmenter.state_dict().keys())))


def golden_autoencoder(ref, tag, B, N, seed):
    """Reference autoencoder (models/autoencoder.py:62-125): encoder -> FC + conv decoder -> multi-resolution Chamfer
    loss, eval forward plus the gradient of the loss w.r.t. the predicted clouds.  The nearest-neighbour search is the
    harness's exact flat-L2 stand-in for faiss; everything else is the reference's own code."""
    opt = ref_harness.make_opt(batch_size=B, input_pc_num=N, output_fc_pc_num=256, output_conv_pc_num=1024)
    model = ref.autoencoder.Model(opt)
    synth.fill_state_dict_(model.encoder.state_dict(), seed=seed)
    synth.fill_state_dict_(model.decoder.state_dict(), seed=seed + 1)
    inp = synth.make_inputs(B, N, M=opt.node_num, som_k=opt.som_k, seed=seed, node_kind="som")
    model.set_input(inp["pc"], inp["sn"], inp["label"], inp["node"], inp["node_knn_I"])
    with ref_harness.sorted_topk(), torch.no_grad():
        model.test_model()
    crit = model.chamfer_criteria
    pred = model.predicted_pc.detach().clone().requires_grad_(True)
    loss2 = crit(pred, model.pc)
    loss2.backward()
    save("autoencoder_" + tag, B=B, N=N, seed=seed, pc=inp["pc"], sn=inp["sn"], node=inp["node"], node_knn_I=inp["node_knn_I"],
         feature=model.feature, predicted_pc=model.predicted_pc, conv_pc4=model.decoder.conv_pc4,
         loss=model.loss, loss_chamfer=model.loss_chamfer, loss_chamfer_conv4=model.loss_chamfer_conv4,
         forward_loss=crit.forward_loss, backward_loss=crit.backward_loss, loss_array=crit.loss_array,
         grad_predicted=pred.grad, decoder_keys=np.array(sorted(model.decoder.state_dict().keys())))


def golden_som_update(ref):
    """BatchSOM.batch_update (util/som.py:295-350): batched SOM training iterations from given nodes (the potential-field
    initialiser is bypassed: nodes start from a seeded uniform layout)."""
    B, N = 2, 3000
    g = torch.Generator().manual_seed(77)
    x = torch.rand(B, 3, N, generator=g) * 2 - 1
    node0 = torch.rand(B, 3, 64, generator=g) * 1.6 - 0.8
    s = ref.som.BatchSOM(8, 8, 3, 0, B)
    s.node.copy_(node0)
    s.batch_update(x, s.learning_rate, s.sigma)
    node1 = s.node.clone()
    for it in range(5):
        decay = 1 + 2 * it / 5
        s.batch_update(x, s.learning_rate / decay, s.sigma / decay)
    save("som_update_b2_n3000", x=x, node0=node0, node1=node1, node6=s.node, weighting=s.get_weighting_matrix(0.3))


def main():
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "som_update":
        torch.manual_seed(0)
        golden_som_update(ref_harness.import_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "autoencoder":     # own process: needs the faiss stand-in at import time
        torch.manual_seed(0)
        torch.set_num_threads(8)
        ref = ref_harness.import_reference(with_faiss_shim=True)
        golden_autoencoder(ref, "b2_n1024", B=2, N=1024, seed=401)
        golden_autoencoder(ref, "b2_n5000", B=2, N=5000, seed=402)            # BASELINE configs[3] size: 5000 gt vs 1280 predicted points
        return
    if len(sys.argv) > 1 and sys.argv[1] == "train512":        # the small training fixture alone (as main() below makes it)
        torch.manual_seed(0)
        torch.set_num_threads(8)
        golden_train_step(ref_harness.import_reference(), "b16_n512", B=16, N=512, seed=201)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "train5000":       # own mode: the configs[1] / configs[4] point count, ~1.5 min of CPU
        torch.manual_seed(0)
        torch.set_num_threads(8)
        golden_train_step(ref_harness.import_reference(), "b8_n5000", B=8, N=5000, seed=203)
        return
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = ref_harness.import_reference()
    golden_state_dict_keys(ref)
    golden_index_max(ref)
    golden_query_topk(ref)
    golden_layers(ref)
    golden_classifier(ref, "b2_n256", B=2, N=256, seed=101, node_kind="uniform")
    golden_classifier(ref, "b8_n1024", B=8, N=1024, seed=102, node_kind="som")          # BASELINE configs[0]
    golden_classifier(ref, "b2_n5000", B=2, N=5000, seed=103, node_kind="som")          # configs[1] shape
    golden_train_step(ref, "b16_n512", B=16, N=512, seed=201)   # B=16: BN over 4 samples is too ill-conditioned to compare gradients
    golden_segmenter(ref, "b2_n256", B=2, N=256, seed=301)
    golden_segmenter(ref, "b2_n1024", B=2, N=1024, seed=302)                  # BASELINE configs[2] size
    golden_som_update(ref)
    golden_classifier(ref, "b2_n300_k1_center", B=2, N=300, seed=104, node_kind="uniform", k=1,
                      som_k=5, som_k_type="center")


if __name__ == "__main__":
    main()
