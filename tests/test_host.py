"""CPU-side tests: parameter inventory, C-ABI surface, host logic, the data-parallel bucket reducer
(world_size 2, gloo)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from oracle import vault_oracle as O
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, fill_param, param_entries, synthetic_batch
from vault_amd.train import BucketReducer, linear_schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parameter_counts_match_survey():
    def count(spec):
        return sum(int(np.prod(s)) for _, s, _ in param_entries(spec))
    assert count(VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)) == 245_906_691
    assert count(VaultSpec(vilt=ViltSpec(), lm=LMSpec.bert_base_uncased(), n_classes=3)) == 220_488_963


def test_names_cover_reference_parameter_names():
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_bertweet_b2.npz"))
    ours = {n for n, _, _ in param_entries(VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3))}
    assert {str(n) for n in g["grad_names"]} <= ours


def test_filler_is_deterministic_and_order_free():
    a = fill_param("encoder.layer.3.output.dense.weight", (8, 4), "normal", 0)
    b = fill_param("encoder.layer.3.output.dense.weight", (8, 4), "normal", 0)
    c = fill_param("encoder.layer.3.output.dense.weight", (8, 4), "normal", 1)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert abs(float(fill_param("x", (100000,), "ln_w").mean()) - 1.0) < 1e-3


def test_synthetic_batch_contract():
    spec = VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)
    b = synthetic_batch(spec, 5, seed=1)
    assert b["input_ids"].shape == (5, 40) and b["input_ids"].dtype == np.int64
    assert b["pixel_values"].shape == (5, 3, 384, 384) and b["pixel_values"].dtype == np.float32
    assert (b["input_ids"][b["attention_mask"] == 0] == 1).all()      # pad id
    assert (b["input_ids"][:, 0] == 0).all()
    lens = b["attention_mask"].sum(1)
    assert lens.min() >= 8 and lens.max() <= 40
    assert "token_type_ids" not in b                                   # BERTweet tokenizer returns none


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
def test_library_exports_every_declared_symbol(fmt):
    """Both builds of the library (bf16 / IEEE fp16 operand type, the same sources: vault_amd/build.py) export the one ABI of
    include/vault_hip.h, and each says which build it is."""
    from vault_amd import build
    path = build.VARIANTS[fmt][0]
    if not os.path.exists(path):
        build.build()
    lib = ctypes.CDLL(path)   # torch (imported above) is already in the process
    hdr = open(os.path.join(ROOT, "include", "vault_hip.h")).read()
    names = set(re.findall(r"\b(vault_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    lib.vault_abi_version.restype = ctypes.c_int
    assert lib.vault_abi_version() == 12
    lib.vault_operand_format.restype = ctypes.c_int
    assert lib.vault_operand_format() == ("bf16", "fp16").index(fmt)


def test_ctypes_structures_match_the_c_header(tmp_path):
    """The ctypes mirrors of the argument structs (vault_amd/lib.py, ops.py) have the size and the field offsets gcc
    gives the structs of include/vault_hip.h: the FFI boundary cannot drift silently."""
    import ctypes as C
    import shutil
    import subprocess
    from vault_amd import lib as L, ops, preprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    pairs = {"vault_gemm_args": L.GemmArgs, "vault_ln_fwd_args": ops.LnFwdArgs, "vault_ln_bwd_args": ops.LnBwdArgs,
             "vault_attn_args": ops.AttnArgs, "vault_gather_args": ops.GatherArgs, "vault_head_args": ops.HeadArgs,
             "vault_layer_args": ops.LayerArgs, "vault_layer_bwd_args": ops.LayerBwdArgs,
             "vault_lm_embed_args": ops.LmEmbedArgs, "vault_text_embed_args": ops.TextEmbedArgs,
             "vault_patch_embed_args": ops.PatchEmbedArgs, "vault_head_loss_args": ops.HeadLossArgs,
             "vault_model_dims": ops.ModelDims, "vault_image_desc": preprocess.ImageDesc,
             "vault_wgrad_seg": ops.WgradSeg, "vault_wgrad_grouped_args": ops.WgradGroupedArgs,
             "vault_preprocess_args": preprocess.PreprocessArgs}
    hdr = os.path.join(ROOT, "include", "vault_hip.h")
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{hdr}"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)   # (unknown field names fail to compile)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = dict(l.rsplit(" ", 1) for l in out if l)
    for cname, cls in pairs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vault_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{f} imports the oracle"
                assert "vault_oracle" not in src, f"{f} references the oracle module"


def test_schedule_matches_oracle():
    for s in range(0, 30):
        assert linear_schedule(2e-5, s, 3, 25) == O.linear_schedule_lr(2e-5, s, 3, 25)


def test_model_requires_gpu_and_keeps_reference_signature():
    import inspect
    from vault_amd.models.vault import VaultForTMSC, VaultModel, VaultProcessor
    sig = inspect.signature(VaultModel.__init__)
    assert list(sig.parameters)[1:7] == ["vilt_config", "bert_config", "freeze_lm", "vilt_dropout_prob",
                                         "use_vilt_position_embeddings", "add_pooling_layer"]
    sig = inspect.signature(VaultForTMSC.__init__)
    assert list(sig.parameters)[1:6] == ["vilt_config", "n_classes", "vilt_dropout_prob", "logging_level", "bert_config"]
    sig = inspect.signature(VaultModel.from_pretrained)
    assert list(sig.parameters)[:4] == ["pretrained_vilt", "pretrained_bert", "freeze_lm", "use_vilt_position_embeddings"]
    assert list(inspect.signature(VaultProcessor.from_pretrained).parameters)[:2] == ["vilt_directory", "bert_directory"]
    spec = VaultSpec.tiny(3)
    m = VaultForTMSC(spec.vilt, n_classes=3, bert_config=spec.lm)
    assert set(m.state_dict()) == {n for n, _, _ in param_entries(m.spec)}
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(input_ids=torch.zeros(1, 40, dtype=torch.long), pixel_values=torch.zeros(1, 3, 192, 192))
    with pytest.raises(ValueError):
        m(pixel_values=torch.zeros(1, 3, 192, 192))


def test_itr_head_class_keys_and_itm_checkpoint_adoption(tmp_path):
    """VaultForImageAndTextRetrieval: encoder keys under ``vilt.``, ``rank_output`` head; from_pretrained of an ITM
    pre-training checkpoint takes row 1 of ``itm_score.fc`` for it (ref: vault/models/vault/model.py:375-405)."""
    import json
    from safetensors.torch import save_file
    from vault_amd.models.vault import VaultForImageAndTextRetrieval
    spec = VaultSpec.tiny(1, "roberta")
    m = VaultForImageAndTextRetrieval(spec.vilt, bert_config=spec.lm)
    keys = set(m.state_dict())
    assert {"rank_output.weight", "rank_output.bias", "vilt.pooler.dense.weight", "vilt.embeddings.cls_token",
            "bert.embeddings.word_embeddings.weight"} <= keys
    assert not any(k.startswith(("classifier.", "embeddings.", "encoder.")) for k in keys)
    assert tuple(m.state_dict()["rank_output.weight"].shape) == (1, spec.vilt.hidden_size)
    with pytest.raises(NotImplementedError):
        m(input_ids=torch.zeros(1, 40, dtype=torch.long), pixel_values=torch.zeros(1, 3, 192, 192),
          labels=torch.zeros(1))
    # a base-ViLT style ITM checkpoint: keys under "vilt.", 2-way itm_score head
    d = tmp_path / "vilt-itm"
    d.mkdir()
    v = spec.vilt
    cfg = {f: getattr(v, f) for f in ("vocab_size", "max_position_embeddings", "type_vocab_size", "modality_type_vocab_size",
                                      "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
                                      "layer_norm_eps", "image_size", "patch_size", "num_channels")}
    json.dump(cfg, open(d / "config.json", "w"))
    nolm = VaultForImageAndTextRetrieval(v)          # ViLT-only twin provides tensors of the right shapes
    sd = {k: (t.clone() + 0.5) for k, t in nolm.state_dict().items() if k.startswith("vilt.")}
    sd["itm_score.fc.weight"] = torch.arange(2 * v.hidden_size, dtype=torch.float32).view(2, -1)
    sd["itm_score.fc.bias"] = torch.tensor([3.0, 7.0])
    save_file(sd, str(d / "model.safetensors"))
    loaded = VaultForImageAndTextRetrieval.from_pretrained(str(d))
    got = loaded.state_dict()
    assert torch.equal(got["rank_output.weight"], sd["itm_score.fc.weight"][1:])
    assert torch.equal(got["rank_output.bias"], sd["itm_score.fc.bias"][1:])
    assert torch.equal(got["vilt.pooler.dense.weight"], sd["vilt.pooler.dense.weight"])


def test_from_pretrained_reads_hf_checkpoints_and_exports_key_for_key(tmp_path):
    """checkpoints written by HuggingFace's own ``save_pretrained`` (ViltModel + RobertaModel, safetensors)
    load into the flat / fused parameter layout, and the exported ``state_dict`` loads back into the HF classes key for
    key (ref: vault/models/vault/model.py:92-128)."""
    from transformers import RobertaConfig, RobertaModel, ViltConfig, ViltModel
    from vault_amd.models.vault import VaultForTMSC
    spec = VaultSpec.tiny(3, "roberta")
    v, lm = spec.vilt, spec.lm
    vc = ViltConfig(vocab_size=v.vocab_size, hidden_size=v.hidden_size, num_hidden_layers=v.num_hidden_layers,
                    num_attention_heads=v.num_attention_heads, intermediate_size=v.intermediate_size,
                    image_size=v.image_size, patch_size=v.patch_size, max_position_embeddings=v.max_position_embeddings)
    rc = RobertaConfig(vocab_size=lm.vocab_size, max_position_embeddings=lm.max_position_embeddings, type_vocab_size=1,
                       hidden_size=lm.hidden_size, num_hidden_layers=lm.num_hidden_layers,
                       num_attention_heads=lm.num_attention_heads, intermediate_size=lm.intermediate_size,
                       layer_norm_eps=lm.layer_norm_eps, pad_token_id=1)
    torch.manual_seed(0)
    hv, hb = ViltModel(vc), RobertaModel(rc, add_pooling_layer=False)
    hv.save_pretrained(str(tmp_path / "vilt")); hb.save_pretrained(str(tmp_path / "bert"))
    m = VaultForTMSC.from_pretrained(str(tmp_path / "vilt"), str(tmp_path / "bert"), n_classes=3)
    ours = m.state_dict()
    skip = lambda k: k.endswith("position_ids") or k.endswith("embeddings.token_type_ids")   # noqa: E731
    n = 0
    for k, t in hv.state_dict().items():
        if not skip(k):
            assert torch.equal(ours[k], t), k
            n += 1
    for k, t in hb.state_dict().items():
        if not skip(k):
            assert torch.equal(ours["bert." + k], t), k
            n += 1
    assert n == len(ours) - 2                                    # everything but the fresh TMSC classifier
    # export: back into fresh HF modules, nothing missing but buffers, nothing unexpected
    hv2, hb2 = ViltModel(vc), RobertaModel(rc, add_pooling_layer=False)
    r = hv2.load_state_dict({k: t for k, t in ours.items() if not k.startswith(("bert.", "classifier."))}, strict=False)
    assert not r.unexpected_keys and all(skip(k) for k in r.missing_keys)
    r = hb2.load_state_dict({k[5:]: t for k, t in ours.items() if k.startswith("bert.")}, strict=False)
    assert not r.unexpected_keys and all(skip(k) for k in r.missing_keys)
    assert torch.equal(hv2.state_dict()["encoder.layer.1.output.dense.weight"], hv.state_dict()["encoder.layer.1.output.dense.weight"])


def test_evaluation_metrics_match_sklearn():
    from sklearn.metrics import precision_recall_fscore_support
    from vault_amd.train import evaluation_metrics
    rng = np.random.default_rng(0)
    for ncls in (2, 3, 6):
        t = rng.integers(0, ncls, size=200)
        p = np.where(rng.random(200) < 0.6, t, rng.integers(0, ncls, size=200))
        if ncls == 6:
            p[p == 5] = 0          # a class that is never predicted, and one (4) that never occurs in the truth
            t[t == 4] = 1
        m = evaluation_metrics(t.tolist(), p.tolist())
        _, _, f1, _ = precision_recall_fscore_support(t, p, average="macro", zero_division=0)
        assert abs(m["macro_f1_score"] - f1) < 1e-12 and abs(m["eval_accuracy"] - float(np.mean(t == p))) < 1e-12


# ---- data parallel: bucketed gradient exchange over gloo, world_size 2 ---------------------------------
class HostKernels:
    """Stand-ins (host tensors, torch ops) for the exchange's device kernels (csrc/exchange.hip), so that the bucket /
    sparse-table / wire logic of BucketReducer can run over gloo on the CPU.  Test harness only; the device kernels are
    compared with the same restatements in tests/test_gpu_train.py."""

    @staticmethod
    def narrow(src, dst, n):
        dst[:n].copy_(src[:n].to(torch.bfloat16))

    @staticmethod
    def widen(src, dst, n):
        dst[:n].copy_(src[:n].float())

    @staticmethod
    def sum_chunks(src, n_src, chunk, out):
        acc = torch.zeros(chunk, dtype=torch.float32)
        for k in range(n_src):
            acc += src[k * chunk:(k + 1) * chunk].float()
        out[:chunk].copy_(acc.to(torch.bfloat16))

    @staticmethod
    def rows_union(keys, n, V, flags, uniq, count):
        k = keys[:n]
        u = torch.unique(k[(k >= 0) & (k < V)])
        uniq[:u.numel()].copy_(u)
        count[0] = u.numel()
        count[1] = int((((k < 0) | (k >= V)) & (k != -1)).sum())

    @staticmethod
    def rows_gather(table, idx, n_rows, H, out):
        out[:n_rows * H].view(n_rows, H).copy_(table.view(-1, H)[idx[:n_rows]])

    @staticmethod
    def rows_scatter(src, idx, n_rows, H, table):
        table.view(-1, H)[idx[:n_rows]] = src[:n_rows * H].view(n_rows, H)


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vault_amd.train import SparseTable
    n = 10_000
    stage_lo = {"head": 9000, "vilt1": 6000, "vilt0": 3000, "vilt_embed": 2500, "lm1": 1500, "lm0": 400, "lm_embed": 0}
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = BucketReducer(g, stage_lo, "lm_embed", bucket_elems=2500, dist=dist, kernels=HostKernels)
    launched = []
    for tag in ["head", "vilt1", "vilt0", "vilt_embed", "lm1", "lm0", "lm_embed"]:
        red.on_stage(tag)
        launched = list(red.launched)
    red.finish()
    expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = {"normal order": bool(torch.equal(g, expect))}
    # second step: the lowest stage arrives BEFORE the stage above it (embedding backward ahead of the last group's deferred
    # weight gradients): its range is reduced at once, the stage above closes the gap
    g.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
    order = []
    for tag in ["head", "vilt1", "vilt0", "vilt_embed", "lm1", "lm_embed", "lm0"]:
        red.on_stage(tag)
        order = list(red.launched)
    x = red.finish_upper()
    ok["early bottom: ranges"] = order[-2] == (0, 400) and order[-1][0] == 400 and x == order[-1][1]
    red.finish()
    ok["early bottom: sum"] = bool(torch.equal(g, expect))
    cover2 = sorted(order)
    ok["early bottom: cover"] = cover2[0][0] == 0 and cover2[-1][1] == n and all(a[1] == b[0] for a, b in zip(cover2, cover2[1:]))
    # ranges are contiguous, descending and cover [0, n) exactly once
    cover = sorted(launched)
    ok["cover"] = cover[0][0] == 0 and cover[-1][1] == n and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    ok["bucket size"] = all(hi - lo >= 2500 for lo, hi in launched[:-1])

    # a bucket larger than the whole buffer, stages in the normal order: everything is pending when the lowest stage
    # arrives - with and without a language model below the ViLT embeddings (regression: the "lowest stage arrived early"
    # branch must not fire here and strand [above_last, hi))
    for last, tags in (("lm_embed", ["head", "vilt1", "vilt0", "vilt_embed", "lm1", "lm0", "lm_embed"]),
                       ("vilt_embed", ["head", "vilt1", "vilt0", "vilt_embed"])):
        lo_map = {t: stage_lo[t] for t in tags}
        if last == "vilt_embed":
            lo_map = {t: v - 2500 for t, v in lo_map.items()}
        m = n if last == "lm_embed" else n - 2500
        g2 = torch.arange(m, dtype=torch.float32) * (rank + 1)
        r2 = BucketReducer(g2, lo_map, last, bucket_elems=10 * n, dist=dist, kernels=HostKernels)
        for rep in range(2):
            for tag in tags:
                r2.on_stage(tag)
            ok[f"huge bucket {last} x{rep}: one launch"] = r2.launched == [(0, m)]
            xx = r2.finish_upper()
            r2.finish()
            ok[f"huge bucket {last} x{rep}: upper"] = xx == m
        ok[f"huge bucket {last}: sum"] = bool(torch.equal(g2, torch.arange(m, dtype=torch.float32) * 6.0))   # (3x, then 2x)
    # per-layer LM order with lm0 still pending below the bucket size when lm_embed arrives in the normal order
    g3 = torch.ones(n) * (rank + 1)
    r3 = BucketReducer(g3, stage_lo, "lm_embed", bucket_elems=3000, dist=dist, kernels=HostKernels)
    for tag in ["head", "vilt1", "vilt0", "vilt_embed", "lm1", "lm0", "lm_embed"]:
        r3.on_stage(tag)
    r3.finish()
    ok["pending lm0"] = bool(torch.equal(g3, torch.ones(n) * 3))

    # bf16 wire: reduce-scatter + all-gather, f32 accumulation of the bf16 images in rank order, one rounding
    vals = [torch.sin(torch.arange(n, dtype=torch.float32) * 0.37 + r) * (1.0 + r) for r in range(world)]
    g4 = vals[rank].clone()
    r4 = BucketReducer(g4, stage_lo, "lm_embed", bucket_elems=2500, dist=dist, wire="bf16", kernels=HostKernels)
    r4.WIRE_PIECE = 4096                                   # several rounds per range, ragged last one
    for tag in ["head", "vilt1", "vilt0", "vilt_embed", "lm1", "lm_embed", "lm0"]:
        r4.on_stage(tag)
    r4.finish()
    want = sum(v.to(torch.bfloat16).float() for v in vals).to(torch.bfloat16).float()
    ok["bf16 wire"] = bool(torch.equal(g4, want))
    exact = sum(vals)
    ok["bf16 wire: close to the f32 sum"] = float((g4 - exact).abs().max()) < 2.0 ** -7 * float(exact.abs().max())

    # row-sparse table: rows [0, 20) x 50 at the bottom of the buffer, each rank touches its own ids
    H, V = 50, 20
    sp = SparseTable(0, V, H)
    lo5 = {"head": 9000, "vilt_embed": 2500, "lm0": 1000, "lm_embed": 0}
    for wire in ("fp32", "bf16"):
        ids = torch.tensor([[3, 7, 7, 1], [19, 3, 0, 1]][rank], dtype=torch.int64)
        g5 = torch.zeros(n)
        g5[1000:] = torch.arange(n - 1000, dtype=torch.float32) * (rank + 1)
        tab = g5[:V * H].view(V, H)
        tab[ids] = torch.arange(H, dtype=torch.float32) + 10.0 * (rank + 1)          # only the touched rows are non-zero
        dense = g5.clone()
        dist.all_reduce(dense)
        r5 = BucketReducer(g5, lo5, "lm_embed", bucket_elems=2500, dist=dist, wire=wire, sparse=sp, kernels=HostKernels)
        for rep, tags in enumerate((["head", "vilt_embed", "lm0", "lm_embed"], ["head", "vilt_embed", "lm_embed", "lm0"])):
            if rep:
                g5.zero_()
                g5[1000:] = torch.arange(n - 1000, dtype=torch.float32) * (rank + 1)
                tab[ids] = torch.arange(H, dtype=torch.float32) + 10.0 * (rank + 1)
            r5.begin_step(ids)
            for tag in tags:
                r5.on_stage(tag)
            r5.finish()
            if wire == "fp32":
                ok[f"sparse table {wire} x{rep}"] = bool(torch.equal(g5, dense))
            else:
                ok[f"sparse table {wire} x{rep}"] = float((g5 - dense).abs().max()) <= 2.0 ** -7 * float(dense.abs().max())
            ok[f"sparse table {wire} x{rep}: bytes"] = r5.wire_bytes < (4 if wire == "fp32" else 2) * n
        # a SHORT batch on one rank (fewer token ids than the count agreed on in the first step): padded with -1, same result
        g5.zero_()
        g5[1000:] = torch.arange(n - 1000, dtype=torch.float32) * (rank + 1)
        short = ids[:2] if rank == 1 else ids
        tab[short] = torch.arange(H, dtype=torch.float32) + 10.0 * (rank + 1)
        dense = g5.clone()
        dist.all_reduce(dense)
        r5.begin_step(short)
        for tag in ["head", "vilt_embed", "lm0", "lm_embed"]:
            r5.on_stage(tag)
        r5.finish()
        ok[f"sparse table {wire}: short batch"] = (bool(torch.equal(g5, dense)) if wire == "fp32" else
                                                   float((g5 - dense).abs().max()) <= 2.0 ** -7 * float(dense.abs().max()))
        # more ids than agreed on, or a switch to inputs_embeds, on ONE rank: EVERY rank raises when the table is exchanged (the
        # header words travel in the step's key all-gather) - no rank is left waiting in a collective - and steps on afterwards
        for what, bad_keys in (("too many ids", torch.cat([ids, ids])), ("switch to inputs_embeds", None)):
            r5.begin_step(bad_keys if rank == 1 else ids)
            try:
                for tag in ["head", "vilt_embed", "lm0", "lm_embed"]:
                    r5.on_stage(tag)
                ok[f"sparse table {wire}: {what} raises on every rank"] = False
            except RuntimeError as e:
                ok[f"sparse table {wire}: {what} raises on every rank"] = "rank(s)" in str(e) and ("[(1, 8)]" in str(e) or "[1]" in str(e))
            try:
                r5.finish()
            except RuntimeError:
                pass
        # a token id outside the table: its row would be left out of the union - raised when the table is exchanged, and
        # the reducer is usable again afterwards (finish() resets before it raises)
        r5.begin_step(torch.tensor([3, V + 5, 1, 0], dtype=torch.int64))
        try:
            for tag in ["head", "vilt_embed", "lm0", "lm_embed"]:
                r5.on_stage(tag)
            ok[f"sparse table {wire}: out-of-range id"] = False
        except RuntimeError as e:
            ok[f"sparse table {wire}: out-of-range id"] = "outside the table" in str(e)
        try:
            r5.finish()
        except RuntimeError:
            pass
        ok[f"sparse table {wire}: reset after error"] = (r5.hi, r5.bottom, r5.launched) == (n, 0, [])
        # the same error CAUGHT by the caller, who steps on without ever reaching finish() - the frontier of the
        # failed step (upper ranges launched, hi at the lowest bucket) must not make the next step skip its upper ranges
        r5.begin_step(torch.cat([ids, ids]) if rank == 1 else ids)
        try:
            for tag in ["head", "vilt_embed", "lm0", "lm_embed"]:
                r5.on_stage(tag)
            ok[f"sparse table {wire}: error before the caught-error step"] = False
        except RuntimeError:
            ok[f"sparse table {wire}: frontier left behind by the failed step"] = r5.hi < n      # (what begin_step must clear)
        g5.zero_()
        g5[1000:] = torch.arange(n - 1000, dtype=torch.float32) * (rank + 1)
        tab[ids] = torch.arange(H, dtype=torch.float32) + 10.0 * (rank + 1)
        dense = g5.clone()
        dist.all_reduce(dense)
        r5.begin_step(ids)
        for tag in ["head", "vilt_embed", "lm0", "lm_embed"]:
            r5.on_stage(tag)
        r5.finish()
        ok[f"sparse table {wire}: step after a caught error reduces every range"] = (
            bool(torch.equal(g5, dense)) if wire == "fp32" else
            float((g5 - dense).abs().max()) <= 2.0 ** -7 * float(dense.abs().max()))
        ok[f"sparse table {wire}: ... and covers the buffer"] = sorted(r5.launched) == [] and (r5.hi, r5.bottom) == (n, 0)
        # a reducer whose steps carry no token ids on ANY rank (inputs_embeds): the table's gradient is zero everywhere,
        # nothing is exchanged for it
        g6 = torch.zeros(n)
        r6 = BucketReducer(g6, lo5, "lm_embed", bucket_elems=2500, dist=dist, wire=wire, sparse=sp, kernels=HostKernels)
        r6.begin_step(None)
        for tag in ["head", "vilt_embed", "lm0", "lm_embed"]:
            r6.on_stage(tag)
        r6.finish()
        ok[f"sparse table {wire}: no ids"] = float(g6.abs().max()) == 0.0
    # VAULT_DP_CHECK_SPARSE (debug): the row-sparse result against a dense all-reduce of the table - passes on a consistent step,
    # raises when a rank holds a gradient row outside the union of the step's token ids (it would stay un-reduced)
    os.environ["VAULT_DP_CHECK_SPARSE"] = "2"
    try:
        for wire in ("fp32", "bf16"):
            ids = torch.tensor([[3, 7, 7, 1], [19, 3, 0, 1]][rank], dtype=torch.int64)
            g8 = torch.zeros(n)
            tab8 = g8[:V * H].view(V, H)
            tab8[ids] = torch.arange(H, dtype=torch.float32) + 10.0 * (rank + 1)
            r8 = BucketReducer(g8, lo5, "lm_embed", bucket_elems=2500, dist=dist, wire=wire, sparse=sp, kernels=HostKernels)
            r8.begin_step(ids)
            for tag in ["head", "vilt_embed", "lm0", "lm_embed"]:
                r8.on_stage(tag)
            r8.finish()
            ok[f"sparse check {wire}: consistent step passes"] = r8.sparse_checks == 1
            g8.zero_()
            tab8[ids] = 1.0
            if rank == 1:
                tab8[11] = 5.0                       # a row no rank's ids name
            r8.begin_step(ids)
            try:
                for tag in ["head", "vilt_embed", "lm0", "lm_embed"]:
                    r8.on_stage(tag)
                ok[f"sparse check {wire}: stray row raises"] = False
            except RuntimeError as e:
                ok[f"sparse check {wire}: stray row raises"] = "differs from the dense" in str(e)
            try:
                r8.finish()
            except RuntimeError:
                pass
            # the steps WITHOUT any touched row (inputs_embeds on every rank; an empty union) are checked too - a second source of
            # gradient on the table would go un-reduced altogether there - and count down like the others
            g9 = torch.zeros(n)
            r9 = BucketReducer(g9, lo5, "lm_embed", bucket_elems=2500, dist=dist, wire=wire, sparse=sp, kernels=HostKernels)
            r9.begin_step(None)
            for tag in ["head", "vilt_embed", "lm0", "lm_embed"]:
                r9.on_stage(tag)
            r9.finish()
            ok[f"sparse check {wire}: no-ids step is checked"] = r9.sparse_checks == 1 and r9._check_sparse_left == 1
            if rank == 1:
                g9[:V * H].view(V, H)[4] = 2.0
            r9.begin_step(None)
            try:
                for tag in ["head", "vilt_embed", "lm0", "lm_embed"]:
                    r9.on_stage(tag)
                ok[f"sparse check {wire}: stray row in a no-ids step raises"] = False
            except RuntimeError as e:
                ok[f"sparse check {wire}: stray row in a no-ids step raises"] = "differs from the dense" in str(e)
            try:
                r9.finish()
            except RuntimeError:
                pass
    finally:
        del os.environ["VAULT_DP_CHECK_SPARSE"]
    # ranks that disagree on the token count of the first step: every rank raises instead of hanging in the all-gather
    g7 = torch.zeros(n)
    r7 = BucketReducer(g7, lo5, "lm_embed", bucket_elems=2500, dist=dist, sparse=sp, kernels=HostKernels)
    try:
        r7.begin_step(torch.zeros(4 + rank, dtype=torch.int64))
        ok["first-step mismatch raises"] = False
    except RuntimeError as e:
        ok["first-step mismatch raises"] = "disagree" in str(e)
    q.put((rank, all(ok.values()), [k for k, v in ok.items() if not v], launched))
    dist.destroy_process_group()


def test_bucket_reducer_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _, _ in res), [bad for _, _, bad, _ in res]
    assert res[0][3] == res[1][3]


def _dp_worker8(rank, world, port, q):
    """BucketReducer at the rank count of the scaling run (8): ragged range sizes, both wires, row-sparse table with
    rank-dependent token ids, two steps on one reducer - every rank ends with the dense all-reduce's result."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vault_amd.train import SparseTable
    try:
        n, H, V = 30_000, 48, 100
        lo = {"head": 29_000, "vilt1": 21_000, "vilt0": 13_000, "vilt_embed": 9_000, "lm1": 7_000, "lm0": V * H, "lm_embed": 0}
        sp = SparseTable(0, V, H)
        ok = {}
        for wire in ("fp32", "bf16"):
            g = torch.zeros(n)
            red = BucketReducer(g, lo, "lm_embed", bucket_elems=6_000, dist=dist, wire=wire, sparse=sp, kernels=HostKernels)
            for step in range(2):
                gen = torch.Generator().manual_seed(1000 * step + rank)
                g.copy_(torch.randn(n, generator=gen))
                ids = torch.randint(0, V, (12,), generator=gen)
                tab = g[:V * H].view(V, H)
                keep = torch.zeros(V, dtype=torch.bool); keep[ids] = True
                tab[~keep] = 0.0                                  # only the rows this rank's ids name carry gradient
                dense = g.clone()
                dist.all_reduce(dense)
                red.begin_step(ids)
                for tag in ("head", "vilt1", "vilt0", "vilt_embed", "lm1", "lm0", "lm_embed"):
                    red.on_stage(tag)
                red.finish()
                if wire == "fp32":
                    ok[f"{wire} step {step}"] = float((g - dense).abs().max()) <= 1e-5 * float(dense.abs().max())
                else:
                    ok[f"{wire} step {step}"] = float((g - dense).abs().max()) <= 2.0 ** -6 * float(dense.abs().max())
                same = [torch.zeros_like(g) for _ in range(world)]
                dist.all_gather(same, g)
                ok[f"{wire} step {step}: replicas identical"] = all(torch.equal(same[0], t) for t in same)
        q.put((rank, all(ok.values()), [k for k, v in ok.items() if not v]))
    finally:
        dist.destroy_process_group()


def test_bucket_reducer_gloo_world8():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_dp_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), [bad for _, _, bad in res]


def test_embedding_surgery_api_and_vault_alias_package():
    """ref model.py:130-149 / 499-509 on the CPU-resident module: resize_token_embeddings keeps the old rows, the
    get -> rewrite -> set sequence of ``integrate_entities_into_model`` (ref: vault/entity_linking.py:133-148) lands in
    the state_dict, renew_classifier swaps the VQA output projection; ``vault.models.vault`` is an import alias."""
    from vault.models.vault import VaultForQuestionAnswering as AliasVQA, VaultForTMSC, VaultProcessor  # noqa: F401
    import vault_amd.models.vault as impl
    assert VaultForTMSC is impl.VaultForTMSC and AliasVQA is impl.VaultForQuestionAnswering
    spec = VaultSpec.tiny(3, "roberta")
    m = VaultForTMSC(spec.vilt, n_classes=3, bert_config=spec.lm)
    name = "bert.embeddings.word_embeddings.weight"
    old = m.state_dict()[name].clone()
    V = old.shape[0]
    emb = m.resize_token_embeddings(V + 5)
    assert emb is m.get_input_embeddings() and tuple(emb.weight.shape) == (V + 5, spec.vilt.hidden_size)
    assert m.spec.lm.vocab_size == V + 5 and torch.equal(m.state_dict()[name][:V], old)
    assert set(m.state_dict()) == {n for n, _, _ in param_entries(m.spec)}
    # the reference's entity integration: clone the table, overwrite the last rows, re-assign, hand back
    ecls = m.get_input_embeddings()
    table = ecls.weight.clone()
    table[-1] = table[[3, 4, 9]].max(0)[0]
    ecls.weight = torch.nn.parameter.Parameter(table)
    m.set_input_embeddings(ecls)
    assert torch.equal(m.state_dict()[name], table.detach())
    assert dict(m.named_parameters())[name] is m._params_by_name[name]
    m.resize_token_embeddings(V)                                   # shrinking keeps the first rows
    assert torch.equal(m.state_dict()[name], old)
    q = AliasVQA(spec.vilt, bert_config=spec.lm, n_classes=7)
    w0 = q.state_dict()["classifier.0.weight"].clone()
    q.renew_classifier(11)
    sd = q.state_dict()
    assert tuple(sd["classifier.3.weight"].shape) == (11, 2 * spec.vilt.hidden_size) and float(sd["classifier.3.bias"].abs().max()) == 0.0
    assert torch.equal(sd["classifier.0.weight"], w0)
    # positional order of the reference's pinned ViltModel.forward: head_mask sits at index 5
    with pytest.raises(NotImplementedError, match="head_mask"):
        m._collect_batch([torch.zeros(1, 40, dtype=torch.long), None, None, torch.zeros(1, 3, 192, 192), None,
                          torch.ones(2, 4)], {})


def _bench_refusal():
    """`python bench.py --gpus N` with N above the node's device count: the launcher parent counts devices from sysfs (no HIP
    call in the process that would fork the ranks), prints the refusal and exits 2 - it never measures fewer ranks than asked."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    have = bench.count_gpus_sysfs()
    n = max(2, have + 1)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert f"--gpus {n} needs {n} devices, this node shows {have}" in r.stderr and "refusing" in r.stderr
    assert r.stdout.strip() == ""            # no JSON line: nothing was measured
    return have


def test_bench_refuses_more_ranks_than_devices():
    _bench_refusal()


def test_sysfs_device_count_honours_visibility_variables(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    have = bench.count_gpus_sysfs()
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.count_gpus_sysfs() == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.count_gpus_sysfs() == min(have, 1)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1,2")
    assert bench.count_gpus_sysfs() == min(have, 1)


def test_operand_format_is_a_context_variable_per_thread():
    """the current operand format (which of the two libraries a launch goes to) must not leak into other host
    threads - a prefetch / preprocessing thread that launches while an fp16 engine is inside its context stays on its own
    default."""
    import threading
    from vault_amd import ops
    seen = []
    with ops.operand_format("fp16"):
        t = threading.Thread(target=lambda: seen.append(ops.current_format()))
        t.start(); t.join()
        assert ops.current_format() == "fp16"
        with ops.operand_format("bf16"):
            assert ops.current_format() == "bf16"
        assert ops.current_format() == "fp16"
    assert seen == ["bf16"] and ops.current_format() == "bf16"


def test_library_override_switches_load_that_file_or_raise():
    """VAULT_HIP_LIB / VAULT_HIP_LIB_F16 (same-box A/B of two builds of the library) name the file the loader opens - there is
    still no fallback: a path that does not exist raises instead of quietly loading the in-tree build."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from vault_amd import lib\n"
            "for fmt in ('bf16', 'fp16'):\n"
            "    try:\n"
            "        lib.load(fmt); print('loaded', fmt)\n"
            "    except RuntimeError as e:\n"
            "        print('raised', fmt, 'not found' in str(e))\n") % ROOT
    env = dict(os.environ, VAULT_HIP_LIB="/nonexistent/libvault_hip.so", VAULT_HIP_LIB_F16="/nonexistent/libvault_hip_f16.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300).stdout
    assert "raised bf16 True" in out and "raised fp16 True" in out, out
    env = dict(os.environ, VAULT_HIP_LIB=os.path.join(ROOT, "vault_amd", "libvault_hip.so"))
    env.pop("VAULT_HIP_LIB_F16", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300).stdout
    assert "loaded bf16" in out and "loaded fp16" in out, out


def test_every_environment_switch_of_the_product_is_named_by_a_test():
    """no kept-but-untested variants - every VAULT_* environment variable the product reads is exercised by
    a test of this directory (compile-time development macros of csrc/ are not environment switches)."""
    import glob
    import re
    read = set()
    for f in glob.glob(os.path.join(ROOT, "vault_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(ROOT, "vault_amd", "csrc", "*.h*")):
        src = open(f).read()
        read |= set(re.findall(r'environ(?:\.get)?\(\s*"(VAULT_[A-Z0-9_]+)"', src)) | set(re.findall(r'environ\[\s*"(VAULT_[A-Z0-9_]+)"', src))
        read |= set(re.findall(r'getenv\(\s*"(VAULT_[A-Z0-9_]+)"', src))
    tests = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "tests", "test_*.py")))
    assert read, "the scan found no switch at all: the patterns are stale"
    missing = sorted(v for v in read if v not in tests)
    assert not missing, missing


def test_isa_check_finds_an_unpadded_sgpr_reload_in_front_of_vmem():
    """vault_amd/isa_check.py (run by the build over every kernel file): a v_readlane_b32 into the SGPR base of a vector-memory
    instruction needs 5 wait states in between; hipcc does not provide them in front of inline asm."""
    from vault_amd.isa_check import sgpr_vmem_hazards
    def kern(name, mid):
        return (f"{name}:\n\ts_load_dwordx2 s[0:1], s[4:5], 0x0\n\tv_readlane_b32 s66, v114, 28\n\tv_readlane_b32 s67, v114, 29\n"
                + mid + "\tglobal_load_dwordx4 v[52:55], v40, s[66:67]\n\ts_endpgm\n.Lfunc_end0:\n")
    bad = kern("_Z3badv", "")
    padded = kern("_Z6paddedv", "\ts_nop 4\n")
    far = kern("_Z3farv", "".join(f"\tv_mov_b32_e32 v{i}, 0\n" for i in range(5)))
    other = kern("_Z5otherv", "").replace("s[66:67]", "s[10:11]")
    found = sgpr_vmem_hazards(bad + padded + far + other)
    assert len(found) == 1 and found[0].startswith("_Z3badv") and "0 wait states" in found[0]
    assert sgpr_vmem_hazards(bad, "padded") == []


def test_isa_check_reports_spilling_kernels():
    """vault_amd/isa_check.py spilling_kernels: the build lists register spills in gemm256.hip / gemm8w.hip (build.py NO_SPILL)."""
    from vault_amd.isa_check import spilling_kernels
    from vault_amd import build
    meta = ("amdhsa.kernels:\n  - .agpr_count: 0\n    .name:           _Z1av\n    .sgpr_count: 10\n    .vgpr_count: 64\n"
            "    .vgpr_spill_count: 0\n  - .agpr_count: 128\n    .name:           _Z1bv\n    .sgpr_count: 90\n    .vgpr_count: 256\n"
            "    .vgpr_spill_count: 62\n")
    assert spilling_kernels(meta) == ["_Z1bv: 62 spilled VGPRs"]
    assert set(build.NO_SPILL) == {"gemm256.hip", "gemm256_dyn.hip", "gemm8w.hip"}


def test_build_refuses_spills_in_the_hot_gemm_kernels_and_the_shipped_ones_have_none():
    """a spill in a HOT instantiation is a build failure (build.hot_spills over NO_SPILL_KERNELS) - every
    8-wave kernel and the static ring forms of the step (weight gradients <1,1,5,4>, data gradients <0,1,0,3|2>, residual
    forwards <0,0,3,3|2>); the dynamic-scheduler twins (gemm256_dyn.hip, data-parallel steps) and cold forms are reported only.
    The spill lists the last build left beside the objects are empty for the hot set (both operand formats)."""
    from vault_amd import build
    hot = ["_ZN12_GLOBAL__N_114gemm256_kernelILi1ELi1ELi5ELi4ELb0EEEv10GemmParams: 2 spilled VGPRs",
           "_ZN12_GLOBAL__N_114gemm256_kernelILi0ELi1ELi0ELi3ELb0EEEv10GemmParams: 1 spilled VGPRs",
           "_ZN12_GLOBAL__N_114gemm256_kernelILi0ELi0ELi3ELi2ELb0EEEv10GemmParams: 1 spilled VGPRs",
           "_ZN12_GLOBAL__N_113gemm8w_kernelILi7ELi4ELb1ELb0EEEv10GemmParams: 4 spilled VGPRs"]
    cold = ["_ZN12_GLOBAL__N_114gemm256_kernelILi1ELi1ELi5ELi4ELb1EEEv10GemmParams: 7 spilled VGPRs",
            "_ZN12_GLOBAL__N_114gemm256_kernelILi0ELi0ELi4ELi4ELb0EEEv10GemmParams: 6 spilled VGPRs"]
    assert build.hot_spills(hot + cold) == hot
    if not all(os.path.exists(os.path.join(od, "gemm256.spills.txt")) for _, od, _ in build.VARIANTS.values()):
        build.build()        # (digest-cached: compiles only what is stale)
    for _, objdir, _ in build.VARIANTS.values():
        for f in build.NO_SPILL:
            lst = os.path.join(objdir, f[:-4] + ".spills.txt")
            assert os.path.exists(lst), f"{lst}: run vault_amd.build first (tests/conftest.py builds the library)"
            assert build.hot_spills(open(lst).read().split("\n")) == []
        # the static translation unit of the ring kernel and the 8-wave kernel: no spill at all
        for f in ("gemm256", "gemm8w"):
            assert open(os.path.join(objdir, f + ".spills.txt")).read().strip() == ""


def test_isa_check_sees_carry_out_writers_single_sgpr_operands_and_branch_targets():
    """the hazard scan also finds an SGPR written as the SECOND operand of a VALU instruction (carry-out forms), a
    single-register SGPR operand of a vector-memory instruction (buffer soffset), and a writer that reaches the access through
    a branch into a label inside the window; an unconditional branch in front of a label cuts the fall-through path."""
    from vault_amd.isa_check import sgpr_vmem_hazards
    def kern(name, pre, vm="global_load_dwordx4 v[52:55], v40, s[66:67]"):
        return f"{name}:\n\ts_load_dwordx2 s[0:1], s[4:5], 0x0\n{pre}\t{vm}\n\ts_endpgm\n.Lfunc_end0:\n"
    assert len(sgpr_vmem_hazards(kern("_Z1bv", "\tv_add_co_u32_e64 v1, s[66:67], v2, v3\n"))) == 1
    assert len(sgpr_vmem_hazards(kern("_Z1cv", "\tv_readfirstlane_b32 s9, v3\n", "buffer_load_dword v1, v2, s[12:15], s9 offen"))) == 1
    assert sgpr_vmem_hazards(kern("_Z1cw", "\tv_readfirstlane_b32 s8, v3\n", "buffer_load_dword v1, v2, s[12:15], s9 offen")) == []
    six = "".join("\tv_mov_b32 v1, 0\n" for _ in range(6))
    through_branch = "\tv_readlane_b32 s66, v114, 28\n\ts_branch .LBB0_2\n.LBB0_1:\n" + six + ".LBB0_2:\n"
    found = sgpr_vmem_hazards(kern("_Z1dv", through_branch))
    assert len(found) == 1 and "1 wait states" in found[0]
    # the same writer behind an unconditional branch that does NOT lead to the access: the label's fall-through is dead
    dead = "\tv_readlane_b32 s66, v114, 28\n\ts_branch .LBB0_9\n.LBB0_2:\n"
    assert sgpr_vmem_hazards(kern("_Z1ev", dead)) == []
    assert sgpr_vmem_hazards(kern("_Z1fv", "\tv_readlane_b32 s66, v114, 28\n\ts_nop 4\n")) == []
