"""Teacher-forced whole-loop parity on the GPU.

The HIP engine decodes with its own bf16 backbone; every iteration's logits are recorded and replayed into the CPU
oracle loop (oracle/loop.py), which draws its noise from an identically seeded device generator.  Every decision
(window ids, sampled ids, accept length, corrected ids, RNG stream position) must then be identical, so the two
token sequences must be identical.  Used by tests/test_gpu_loop.py and __graft_entry__.smoke().
"""
import torch

from oracle import loop as OL
from oracle import sjd_oracle as O


class _Recorder:
    def __init__(self):
        self.items = []

    def __call__(self, d):
        self.items.append(dict(
            first=d["first"], n_rows=d["n_rows"], logits_c=d["logits_c"].float().cpu().numpy().copy(),
            logits_u=None if d["logits_u"] is None else d["logits_u"].float().cpu().numpy().copy(),
            noise=d["noise"].cpu().numpy().copy(), rs=d["rs"].cpu().numpy().copy(),
            noise2=d["noise2"].cpu().numpy().copy(), use_cfg=bool(d["use_cfg"])))


class _StreamRecorder:
    """_Recorder for long decodes: a bounded queue between the engine (producer, through its hook) and the oracle loop running in a
    second thread, so that a whole 768px image at the 7B shapes (~1000 iterations x 17 MB of logits and noise) never sits in host memory."""

    def __init__(self, depth=4):
        import queue
        self.q = queue.Queue(maxsize=depth)
        self.items = self

    def __call__(self, d):
        self.q.put(dict(
            first=d["first"], n_rows=d["n_rows"], logits_c=d["logits_c"].float().cpu().numpy().copy(),
            logits_u=None if d["logits_u"] is None else d["logits_u"].float().cpu().numpy().copy(),
            noise=d["noise"].cpu().numpy().copy(), rs=d["rs"].cpu().numpy().copy(),
            noise2=d["noise2"].cpu().numpy().copy(), use_cfg=bool(d["use_cfg"])))

    def close(self):
        self.q.put(None)

    def __iter__(self):
        while True:
            r = self.q.get()
            if r is None:
                return
            yield r

    def drain(self):
        for _ in self:
            pass


def _replay_while_decoding(rec, decode_fn, prompt, rules_fn, cfg, V, no_cfg_fn=None, device="cuda"):
    """run `decode_fn()` (the engine, feeding `rec`) and the oracle replay concurrently; returns (engine result, oracle result)"""
    import threading
    out = {}

    def oracle():
        try:
            O.set_threads(8)          # omp_set_num_threads is per calling thread: without it this thread's regions open one thread per host core
            out["ref"] = _replay(rec, prompt, rules_fn, cfg, V, no_cfg_fn=no_cfg_fn, device=device)
            leftover = sum(1 for _ in rec)
            if leftover:
                out["err"] = AssertionError(f"the engine ran {leftover} more iterations than the oracle")
        except BaseException as e:          # keep consuming: the engine must not block on a full queue
            out["err"] = e
            rec.drain()

    th = threading.Thread(target=oracle)
    th.start()
    try:
        out["eng"] = decode_fn()
    finally:
        rec.close()
        th.join()
    if "err" in out:
        raise out["err"]
    return out["eng"], out["ref"]


def _replay(rec, prompt, rules_fn, cfg, V, no_cfg_fn=None, device="cuda", grid_fn=None):
    it = iter(rec.items)
    state = {"i": -1}

    def fwd(win, kv_len):
        r = next(it)
        state["i"] += 1
        state["cur"] = r
        return r["logits_c"], r["logits_u"]

    checks = {"noise": 0}

    def hook(kind, d):
        r = state["cur"]
        if kind == "sampled":
            assert (d["logits_u"] is not None) == r["use_cfg"], "CFG on/off decision differs"
            assert (d["noise"].cpu().numpy() == r["noise"]).all(), "Exp(1) noise stream diverged"
            checks["noise"] += 1
        else:
            assert (d["rs"].cpu().numpy() == r["rs"]).all(), "uniform stream diverged"
            assert (d["noise2"].cpu().numpy() == r["noise2"]).all(), "residual noise stream diverged"

    seq, tr = OL.run(prompt, fwd, rules_fn, cfg, V, no_cfg_fn=no_cfg_fn, noise_device=device, hook=hook, grid_fn=grid_fn)
    return seq, tr, checks


def _loop_cfg(c):
    return OL.LoopConfig(jacobi_loop_interval_l=c.jacobi_loop_interval_l, jacobi_loop_interval_r=c.jacobi_loop_interval_r,
                         max_num_new_tokens=c.max_num_new_tokens, guidance_scale=c.guidance_scale, seed=c.seed,
                         do_cfg=c.do_cfg, prefix_token_sampler_scheme=c.prefix_token_sampler_scheme,
                         max_length=c.max_length, eos_token_ids=c.eos_token_ids, multi_token_init_scheme=c.multi_token_init_scheme,
                         do_sample=getattr(c, "do_sample", True))


@torch.no_grad()
def teacher_forced_llamagen_check(device="cuda:0", latent=16, window=16, seed=7, scheme="speculative_jacobi",
                                  embed_token_scale=0.25, top_k=1000, cfg_scale=4.0, dtype=torch.bfloat16,
                                  use_graph=False, top_p=1.0):
    import sjd_amd.ops as ops
    from sjd_amd.engine import SJDEngine, SJDConfig, WindowSpec
    from sjd_amd.grammar import TopKTopPGrammar
    from tests.helpers import make_llamagen, llamagen_prefill_sample
    args = dict(dim=128, n_layer=2, n_head=2, vocab_size=16384, block_size=latent * latent, cls_token_num=1,
                model_type="c2i", num_classes=1000)
    model = make_llamagen(args, 17, embed_token_scale, ops.HipWindowAttention(n_split=2), dtype=dtype, device=device)
    T, N = 1, latent * latent
    s_max = ((T + N + 64 + 31) // 32) * 32
    model.setup_cache(batch=2, s_max=s_max)
    cond = torch.tensor([207, model.num_classes], device=device)
    zeros = torch.zeros(2, dtype=torch.int32, device=device)
    logits = model.forward_embeds(model.embed_condition(cond), torch.zeros(2, 1, dtype=torch.long, device=device), 0, zeros)
    torch.manual_seed(seed)
    first = int(llamagen_prefill_sample(logits.float().cpu(), cfg_scale, 1.0, top_k, top_p)[0, 0])
    cfg = SJDConfig(jacobi_loop_interval_l=1, jacobi_loop_interval_r=N - window - 2, max_num_new_tokens=window,
                    guidance_scale=cfg_scale, seed=seed, prefix_token_sampler_scheme=scheme, max_length=N)
    spec = WindowSpec(first_tokens=torch.tensor([[first], [first]], device=device),
                      first_positions=torch.full((2, 1), T, dtype=torch.long, device=device), key_start=zeros,
                      pos_offset=torch.zeros(2, dtype=torch.long), kv_base=T)
    eng = SJDEngine(model, 16384, device, max_window=window, use_graph=use_graph)
    rec = _Recorder()
    eng.hook = rec
    seq, stats = eng.decode([first], spec, TopKTopPGrammar(top_k, top_p), cfg)
    seq_ref, tr, checks = _replay(rec, [first], lambda c, n: O.llamagen_rules(c, n, top_k, top_p), _loop_cfg(cfg), 16384,
                                  device=device)
    assert seq == seq_ref, "token sequences differ"
    assert stats.matched == tr.matched and stats.nfe == len(tr.matched)
    return dict(tokens=len(seq) - 1, nfe=stats.nfe, tok_per_step=round((len(seq) - 1) / stats.nfe, 3),
                accepted_hist=sorted(set(stats.matched)), noise_checks=checks["noise"])


@torch.no_grad()
def teacher_forced_lumina_check(device="cuda:0", hg=4, wg=4, window=16, seed=3, scheme="speculative_jacobi", P=12,
                                embed_token_scale=0.25, kv_heads=4, l=3, r=None, dtype=torch.bfloat16,
                                use_graph=False, fused=True, gemm="torch", fp8_kv=False, init_scheme="random", temperature=1.0, top_p=None,
                                do_sample=True, uncond_start=None, eos=(8196,)):
    """uncond_start: first prompt token the uncond row sees (default P - 1, the SJD sampler's; P - 3 = the AR baseline's context from the
    image-start token on, IS:62-63)"""
    import sjd_amd.ops as ops
    import sjd_amd.synthetic as synthetic
    from sjd_amd.engine import SJDEngine, SJDConfig, WindowSpec
    from sjd_amd.grammar import LuminaGrammar
    from tests.helpers import make_chameleon
    V = 9216
    conf = dict(vocab_size=V, hidden_size=512, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=kv_heads, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0)
    model = make_chameleon(conf, 23, embed_token_scale, ops.HipWindowAttention(n_split=2), dtype=dtype, device=device)
    if fused:
        model.enable_fused(ops, gemm=gemm)
    prompt = torch.cat([synthetic.synthetic_prompt(P - 3, seed, lo=8900, hi=9200),
                        torch.tensor([[8197, 8804 + hg, 8804 + wg]])], dim=1)
    n_img = (2 * wg + 1) * 2 * hg
    max_len = P + n_img + 1 + 4
    model.setup_cache(batch=2, s_max=((max_len + 64 + 31) // 32) * 32, dtype=ops.FP8 if fp8_kv else None)
    r = r if r is not None else (2 * wg + 1) * 2 * hg - 10
    cfg = SJDConfig(jacobi_loop_interval_l=l, jacobi_loop_interval_r=r, max_num_new_tokens=window, guidance_scale=3.0,
                    seed=seed, prefix_token_sampler_scheme=scheme, max_length=max_len, eos_token_ids=tuple(eos),
                    multi_token_init_scheme=init_scheme, do_sample=do_sample)
    ids = prompt.to(device)
    u0 = P - 1 if uncond_start is None else int(uncond_start)
    spec = WindowSpec(first_tokens=ids.repeat(2, 1),
                      first_positions=torch.stack([torch.arange(P), torch.tensor([1] * u0 + list(range(P - u0)))]).to(device),
                      key_start=torch.tensor([0, u0], dtype=torch.int32),
                      pos_offset=torch.tensor([0, -u0], dtype=torch.long), kv_base=0)
    eng = SJDEngine(model, V, device, max_window=window, use_graph=use_graph)
    rec = _Recorder()
    eng.hook = rec
    gram = LuminaGrammar(2000, 10)
    gram.temperature = float(temperature)          # HF's TemperatureLogitsWarper of the processor list (grammar_from_processors sets it)
    gram.top_p = top_p                             # ... and its TopPLogitsWarper (GenerationConfig.top_p < 1)
    seq, stats = eng.decode(prompt[0].tolist(), spec, gram, cfg)
    seq_ref, tr, checks = _replay(rec, prompt[0].tolist(), O.tempered(lambda c, n: O.lumina_rules(c, n, 2000, 10), temperature, top_p), _loop_cfg(cfg), V,
                                  no_cfg_fn=O.lumina_force_no_cfg, device=device, grid_fn=O.lumina_grid)
    assert seq == seq_ref, "token sequences differ"
    assert stats.matched == tr.matched and stats.nfe == len(tr.matched)
    gen = seq[P:]
    windows = tr.windows
    return dict(tokens=len(gen), nfe=stats.nfe, eol=[i for i, t in enumerate(gen) if t == 8803][:3], last=gen[-1],
                accepted_hist=sorted(set(stats.matched[1:])), noise_checks=checks["noise"], windows=windows)


@torch.no_grad()
def teacher_forced_anole_check(device="cuda:0", img_len=40, window=16, seed=5, P=10, embed_token_scale=0.25, dtype=torch.bfloat16,
                               use_graph=True, gemm="torch", fp8_kv=True):
    """BASELINE config 5 flavour: Chameleon architecture, image-only grammar of the Anole adapter (every window row carries the mask
    of the accepted prefix, reference logit_processor_3dim.py:242-338), no line tokens, fp8 KV cache + fp8-MFMA K1."""
    import sjd_amd.ops as ops
    import sjd_amd.synthetic as synthetic
    from sjd_amd.engine import SJDEngine, SJDConfig, WindowSpec
    from sjd_amd.grammar import AnoleGrammar
    from tests.helpers import make_chameleon
    V = 9216
    conf = dict(vocab_size=V, hidden_size=512, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=4, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0)
    model = make_chameleon(conf, 29, embed_token_scale, ops.HipWindowAttention(n_split=2), dtype=dtype, device=device)
    model.enable_fused(ops, gemm=gemm)
    prompt = torch.cat([synthetic.synthetic_prompt(P - 1, seed, lo=8900, hi=9200), torch.tensor([[8197]])], dim=1)
    max_len = P + img_len + 1
    model.setup_cache(batch=2, s_max=((max_len + 64 + 31) // 32) * 32, dtype=ops.FP8 if fp8_kv else None)
    cfg = SJDConfig(jacobi_loop_interval_l=0, jacobi_loop_interval_r=img_len - window - 2, max_num_new_tokens=window, guidance_scale=3.0,
                    seed=seed, prefix_token_sampler_scheme="speculative_jacobi", max_length=max_len, eos_token_ids=(8196,))
    ids = prompt.to(device)
    spec = WindowSpec(first_tokens=ids.repeat(2, 1),
                      first_positions=torch.stack([torch.arange(P), torch.tensor([1] * (P - 1) + [0])]).to(device),
                      key_start=torch.tensor([0, P - 1], dtype=torch.int32),
                      pos_offset=torch.tensor([0, -(P - 1)], dtype=torch.long), kv_base=0)
    eng = SJDEngine(model, V, device, max_window=window, use_graph=use_graph)
    rec = _Recorder()
    eng.hook = rec
    seq, stats = eng.decode(prompt[0].tolist(), spec, AnoleGrammar(V, P, max_len, img_len), cfg)
    seq_ref, tr, checks = _replay(rec, prompt[0].tolist(), lambda c, n: O.anole_rules(c, n, V, P, max_len, img_len), _loop_cfg(cfg), V,
                                  device=device)
    assert seq == seq_ref, "token sequences differ"
    assert stats.matched == tr.matched and stats.nfe == len(tr.matched)
    gen = seq[P:]
    return dict(tokens=len(gen), nfe=stats.nfe, last=gen[-1], image_ids=all(4 <= t < 8196 for t in gen[:-1]),
                accepted_hist=sorted(set(stats.matched[1:])))


@torch.no_grad()
def teacher_forced_emu3_check(device="cuda:0", H=3, W=5, window=32, seed=5, embed_token_scale=0.4, dtype=torch.float16,
                              use_graph=True, pos_len=9, neg_len=5, gemm="torch", init_scheme="random"):
    """Emu3 flavour (config 3): Llama-style GQA backbone without QK-norm, pos/neg prompts left-padded to a common
    length (pads are hidden keys), EOL/EOF/EOI/EOS grammar, top-k 2048, draft window 32, fp16."""
    import sjd_amd.ops as ops
    import sjd_amd.backbones as BB
    import sjd_amd.synthetic as synthetic
    from sjd_amd.engine import SJDEngine, SJDConfig
    from sjd_amd.frontends import emu3_window_spec
    from sjd_amd.grammar import Emu3Grammar
    V, vis_lo, vis_n = 12288, 3000, 8192
    tok = dict(img_token=200, eoi_token=201, eos_token=202, eol_token=203, eof_token=204, pad_token=205)
    args = BB.ChameleonArgs(vocab_size=V, hidden_size=1024, intermediate_size=512, num_hidden_layers=2, num_attention_heads=8,
                            num_key_value_heads=2, rope_theta=1000000.0, qk_norm=False)
    model = BB.ChameleonBackbone(args, attn=ops.HipWindowAttention(n_split=2)).eval()
    synthetic.fill_state_dict(model, seed=29, embed_token_scale=embed_token_scale)
    model = model.to(device=device, dtype=dtype)
    model.G1_CFG = dict(qkv=(256, 8, True), o=(256, 4, False), gate_up=(512, 8, True), down=(256, 4, False))
    model.enable_fused(ops, gemm=gemm)
    g = torch.Generator().manual_seed(seed)
    pos_ids = torch.randint(300, 2000, (pos_len - 1,), generator=g).tolist() + [tok["img_token"]]
    neg_ids = torch.randint(300, 2000, (neg_len - 1,), generator=g).tolist() + [tok["img_token"]]
    spec = emu3_window_spec(pos_ids, neg_ids, tok["pad_token"], device)
    prompt = spec.first_tokens[0].tolist()
    P = len(prompt)
    n_gen = (W + 1) * H + 3
    max_len = P + n_gen + 2
    model.setup_cache(batch=2, s_max=((max_len + 64 + 31) // 32) * 32)
    cfg = SJDConfig(jacobi_loop_interval_l=1, jacobi_loop_interval_r=(W + 1) * H - 1, max_num_new_tokens=window, guidance_scale=3.0,
                    seed=seed, prefix_token_sampler_scheme="speculative_jacobi", max_length=max_len, eos_token_ids=(tok["eos_token"],),
                    multi_token_init_scheme=init_scheme)
    eng = SJDEngine(model, V, device, max_window=window, use_graph=use_graph)
    rec = _Recorder()
    eng.hook = rec
    seq, stats = eng.decode(prompt, spec, Emu3Grammar(H, W, vis_lo, vis_n, **tok, top_k=2048), cfg)
    rules_fn = lambda c, n: O.emu3_rules(c, n, H, W, vis_lo, vis_n, top_k=2048, **tok)
    seq_ref, tr, checks = _replay(rec, prompt, rules_fn, _loop_cfg(cfg), V, device=device,
                                  grid_fn=lambda c: O.emu3_grid(c, H, W, vis_lo, vis_n, tok["img_token"]))
    assert seq == seq_ref, "token sequences differ"
    assert stats.matched == tr.matched and stats.nfe == len(tr.matched)
    gen = seq[P:]
    return dict(tokens=len(gen), nfe=stats.nfe, gen=gen, tok=tok, W=W, H=H)


@torch.no_grad()
def teacher_forced_batch_check(device="cuda:0", n_prompts=2, hg=4, wg=4, window=16, seed=3, P=(12, 9), embed_token_scale=0.25,
                               dtype=torch.bfloat16, use_graph=True, gemm="sjd", fp8_kv=False, n_slots=None, init_scheme="random", do_sample=True):
    """Several prompts per window forward (SJDBatchEngine): every slot's recorded logits, replayed into the CPU oracle with the slot's
    seed, must give that slot's token sequence and accept lengths -- i.e. sharing the forward changes nothing in any prompt's
    state machine (own window, kv_len, grammar, generators).  n_slots < n_prompts: continuous batching -- a slot that finished its
    image takes the next prompt of the list; every prompt must still decode exactly as its own oracle replay says."""
    import sjd_amd.ops as ops
    import sjd_amd.synthetic as synthetic
    from sjd_amd.engine import SJDConfig, WindowSpec
    from sjd_amd.engine_batch import SJDBatchEngine
    from sjd_amd.grammar import LuminaGrammar
    from tests.helpers import make_chameleon
    V = 9216
    conf = dict(vocab_size=V, hidden_size=512, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=4, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0)
    model = make_chameleon(conf, 23, embed_token_scale, ops.HipWindowAttention(n_split=2), dtype=dtype, device=device)
    wide = (n_slots or n_prompts) * 2 * window > 128
    if wide:        # 129..256 window rows (five to eight prompts per forward, round 5): four-wave workgroups on the uncompressed packing
        model.G1_CFG = dict(qkv=(256, 4, True), o=(128, 4, True), gate_up=(256, 4, True), down=(128, 4, True))
        model.HEAD_CFG = (256, 4, True)
    model.enable_fused(ops, gemm=gemm, compress=False if wide else None)
    n_img = (2 * wg + 1) * 2 * hg
    prompts, specs = [], []
    for i in range(n_prompts):
        Pi = P[i % len(P)]
        pr = torch.cat([synthetic.synthetic_prompt(Pi - 3, seed + 17 * i, lo=8900, hi=9200), torch.tensor([[8197, 8804 + hg, 8804 + wg]])], dim=1)
        prompts.append(pr[0].tolist())
        specs.append(WindowSpec(first_tokens=pr.to(device).repeat(2, 1),
                                first_positions=torch.stack([torch.arange(Pi), torch.tensor([1] * (Pi - 1) + [0])]).to(device),
                                key_start=torch.tensor([0, Pi - 1], dtype=torch.int32),
                                pos_offset=torch.tensor([0, -(Pi - 1)], dtype=torch.long), kv_base=0))
    max_len = max(P) + n_img + 1 + 4
    n_slots = n_slots or n_prompts
    model.setup_cache(batch=2 * n_slots, s_max=((max_len + 64 + 31) // 32) * 32, dtype=ops.FP8 if fp8_kv else None)
    cfg = SJDConfig(jacobi_loop_interval_l=3, jacobi_loop_interval_r=n_img - 10, max_num_new_tokens=window, guidance_scale=3.0,
                    seed=seed, prefix_token_sampler_scheme="speculative_jacobi", max_length=1 << 20, eos_token_ids=(8196,),
                    multi_token_init_scheme=init_scheme, do_sample=do_sample)
    eng = SJDBatchEngine(model, V, device, n_slots, max_window=window, use_graph=use_graph)
    if wide and gemm == "sjd":       # the hand-written projections must be what runs (not the library-GEMM prefill path)
        calls = []
        orig = ops.skinny_gemm
        def spy(x_, *a, **k):
            calls.append(int(x_.shape[0]))
            return orig(x_, *a, **k)
        ops.skinny_gemm = spy
    recs = [_Recorder() for _ in range(n_prompts)]
    eng.hook = lambda i, d: recs[i](d)
    try:
        results = eng.decode_many(prompts, specs, [LuminaGrammar(2000, 10) for _ in range(n_prompts)], cfg)
    finally:
        if wide and gemm == "sjd":
            ops.skinny_gemm = orig
    if wide and gemm == "sjd":
        assert calls and max(calls) == n_slots * 2 * window, "the window forward did not run on G1 at its full row count"
    out = []
    for i, (seq, stats) in enumerate(results):
        c = _loop_cfg(cfg)
        c.seed = cfg.seed + i
        seq_ref, tr, checks = _replay(recs[i], prompts[i], lambda cx, n: O.lumina_rules(cx, n, 2000, 10), c, V,
                                      no_cfg_fn=O.lumina_force_no_cfg, device=device, grid_fn=O.lumina_grid)
        assert seq == seq_ref, f"slot {i}: token sequences differ"
        assert stats.matched == tr.matched, f"slot {i}: accept lengths differ"
        gen = seq[len(prompts[i]):]
        out.append(dict(tokens=len(gen), nfe=stats.nfe, last=gen[-1], max_accept=max(stats.matched[1:])))
    return out


# ------------------------------------------------------------------------------------------------ through the reference API
@torch.no_grad()
def teacher_forced_emu3_api_check(device="cuda:0", H=3, W=5, window=16, seed=5, embed_token_scale=0.4, dtype=torch.float16, pos_len=9,
                                  neg_len=5, gemm="sjd"):
    """The Emu3 flow exactly as reference test_emu3.py:145-169 drives it -- Emu3Processor.build_prefix_constrained_fn -> renew_solver ->
    model.prepare_batch_cfg_model_inputs -> model.generate(pos_ids, GENERATION_CONFIG, logits_processor=, attention_mask=,
    neg_input_ids=) -- on this package's backbone; the engine's logits are replayed into the CPU oracle loop (teacher forcing)."""
    from types import SimpleNamespace as NS
    from transformers import GenerationConfig
    import sjd_amd.ops as ops
    import sjd_amd.backbones as BB
    import sjd_amd.synthetic as synthetic
    from emu3.mllm.processing_emu3 import Emu3Processor
    from scheduler.jacobi_iteration_emu3 import renew_solver
    from tests.helpers import Emu3StubTokenizer
    V, vis_n = 12288, 8192
    tk = Emu3StubTokenizer()
    tok = dict(img_token=200, eoi_token=201, eos_token=202, eol_token=203, eof_token=204, pad_token=205)
    args = BB.ChameleonArgs(vocab_size=V, hidden_size=1024, intermediate_size=512, num_hidden_layers=2, num_attention_heads=8,
                            num_key_value_heads=2, rope_theta=1000000.0, qk_norm=False, max_position_embeddings=256)
    model = BB.ChameleonBackbone(args, attn=ops.HipWindowAttention(n_split=2)).eval()
    synthetic.fill_state_dict(model, seed=29, embed_token_scale=embed_token_scale)
    model = model.to(device=device, dtype=dtype)
    model.G1_CFG = dict(qkv=(256, 8, True), o=(256, 4, False), gate_up=(512, 8, True), down=(256, 4, False))
    model.enable_fused(ops, gemm=gemm)
    model.config = NS(pad_token_id=tok["pad_token"], eos_token_id=tok["eos_token"], image_area=(8 * H) * (8 * W))
    processor = Emu3Processor(None, NS(config=NS(codebook_size=vis_n), spatial_scale_factor=8), tk)
    g = torch.Generator().manual_seed(seed)
    pos_ids = torch.tensor([torch.randint(300, 2000, (pos_len - 1,), generator=g).tolist() + [tok["img_token"]]], device=device)
    neg_ids = torch.tensor([torch.randint(300, 2000, (neg_len - 1,), generator=g).tolist() + [tok["img_token"]]], device=device)
    jac = dict(jacobi_loop_interval_l=1, jacobi_loop_interval_r=(W + 1) * H - 1, max_num_new_tokens=window, guidance_scale=3.0, seed=seed,
               multi_token_init_scheme='random', do_cfg=True, image_top_k=2048, text_top_k=10, prefix_token_sampler_scheme='speculative_jacobi',
               h=H, w=W, neg_inputs=neg_ids, classifier_free_guidance=3.0)
    model, logits_processor = renew_solver(model, processor, **jac)                       # test_emu3.py:145-146
    gc = GenerationConfig(use_cache=True, eos_token_id=tok["eos_token"], pad_token_id=tok["pad_token"], max_new_tokens=40960,
                          do_sample=True, top_k=2048)                                      # test_emu3.py:81-90
    mi = model.prepare_batch_cfg_model_inputs(pos_ids, neg_input_ids=neg_ids, attention_mask=None)   # test_emu3.py:149-155
    assert mi["attention_mask"].shape == (2, pos_len) and mi["pos_input_ids"].shape == (1, pos_len)
    assert mi["attention_mask"][1, :pos_len - neg_len].sum() == 0 and mi["attention_mask"][0].all()
    eng = model._sjd_engine(2, torch.device(device))
    rec = _Recorder()
    eng.hook = rec
    out = model.generate(mi["pos_input_ids"], gc, logits_processor=logits_processor, attention_mask=mi["attention_mask"],
                         neg_input_ids=neg_ids)                                            # test_emu3.py:163-169
    seq = out[0].tolist()
    prompt = mi["pos_input_ids"][0].tolist()
    vis_lo = processor.build_prefix_constrained_fn(H, W).visual_tokens[0]
    cfg = OL.LoopConfig(jacobi_loop_interval_l=1, jacobi_loop_interval_r=(W + 1) * H - 1, max_num_new_tokens=window, guidance_scale=3.0,
                        seed=seed, do_cfg=True, prefix_token_sampler_scheme="speculative_jacobi", max_length=256,
                        eos_token_ids=(tok["eos_token"],))
    rules_fn = lambda c, n: O.emu3_rules(c, n, H, W, vis_lo, vis_n, top_k=2048, **tok)
    seq_ref, tr, checks = _replay(rec, prompt, rules_fn, cfg, V, device=device)
    assert seq == seq_ref, "token sequences differ"
    assert model.last_sjd_stats.matched == tr.matched
    return dict(gen=seq[len(prompt):], tok=tok, W=W, H=H, vis=(vis_lo, vis_n), nfe=model.last_sjd_stats.nfe)


@torch.no_grad()
def teacher_forced_anole_api_check(device="cuda:0", img_len=36, window=16, seed=5, P=10, embed_token_scale=0.25, dtype=torch.bfloat16,
                                   gemm="sjd", fp8_kv=False, mode="image-only", extra_new_tokens=0):
    """The Anole flow as reference model_loader.py:82-108 / 396-411 drives it: renew_pipeline_sampler(model, processor, **kw) then
    model.generate(input_ids, multimodal_generation_mode="image-only", max_new_tokens=L+2, do_sample=True); teacher-forced replay."""
    from types import SimpleNamespace as NS
    import sjd_amd.ops as ops
    import sjd_amd.synthetic as synthetic
    from scheduler.jacobi_iteration_anhole import renew_pipeline_sampler
    from tests.helpers import make_chameleon
    V = 9216
    conf = dict(vocab_size=V, hidden_size=512, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=4, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0)
    model = make_chameleon(conf, 29, embed_token_scale, ops.HipWindowAttention(n_split=2), dtype=dtype, device=device)
    model.G1_CFG = dict(qkv=(256, 8, True), o=(256, 4, False), gate_up=(256, 8, True), down=(256, 4, False))
    model.enable_fused(ops, gemm=gemm)
    kw = dict(jacobi_loop_interval_l=1, jacobi_loop_interval_r=img_len - window - 2, max_num_new_tokens=window, guidance_scale=3.0, seed=seed,
              multi_token_init_scheme='random', do_cfg=True, image_top_k=2000, text_top_k=10, prefix_token_sampler_scheme='speculative_jacobi')
    model = renew_pipeline_sampler(model, NS(image_seq_length=img_len), **kw)               # ML:89-104
    assert model.model.image_seq_length == img_len
    ids = torch.cat([synthetic.synthetic_prompt(P - 1, seed, lo=8900, hi=9200), torch.tensor([[8197]])], dim=1).to(device)
    eng = model._sjd_engine(2, torch.device(device))
    rec = _Recorder()
    eng.hook = rec
    out = model.generate(ids, multimodal_generation_mode=mode, max_new_tokens=img_len + 2 + extra_new_tokens, do_sample=True)   # ML:406-411
    seq = out[0].tolist()
    max_len = P + img_len + 2 + extra_new_tokens
    cfg = OL.LoopConfig(jacobi_loop_interval_l=1, jacobi_loop_interval_r=img_len - window - 2, max_num_new_tokens=window, guidance_scale=3.0,
                        seed=seed, do_cfg=True, prefix_token_sampler_scheme="speculative_jacobi", max_length=max_len, eos_token_ids=())
    seq_ref, tr, checks = _replay(rec, ids[0].tolist(), lambda c, n: O.anole_rules(c, n, V, P, max_len, img_len, top_k=0, mode=mode), cfg, V, device=device)   # no TopK warper in JA:183-232
    assert seq == seq_ref, "token sequences differ"
    assert model.last_sjd_stats.matched == tr.matched
    gen = seq[P:]
    return dict(gen=gen, img_len=img_len, nfe=model.last_sjd_stats.nfe)


@torch.no_grad()
def teacher_forced_real_shape_check(family="lumina7b", device="cuda:0", prompt_len=700, new_tokens=56, seed=11, stream=False):
    """Teacher-forced parity at the REAL architecture shapes and production launch configuration (BASELINE.json configs 2 / 3):
    Lumina-mGPT-7B (32 layers, hidden 4096, 32 heads, V=65536, window 16, G1_CFG, K1 auto-split 4, output head on the grammar's column
    window, hipGraph) or Emu3-Gen-8B (GQA 32/8, V=184622, fp16, window 32, G1_CFG_EMU3, k1_partial_shared).  A prompt of `prompt_len`
    text ids puts the K1 launches into the multi-split regime; ~`new_tokens` tokens are decoded (~25 SJD iterations) and every
    iteration's logits are replayed into the CPU oracle, which must take identical decisions."""
    import sjd_amd.ops as ops
    import sjd_amd.backbones as BB
    import sjd_amd.synthetic as synthetic
    from sjd_amd.engine import SJDEngine, SJDConfig
    from sjd_amd.frontends import lumina_window_spec, lumina_prompt, emu3_window_spec
    from sjd_amd.grammar import LuminaGrammar, Emu3Grammar
    dev = torch.device(device)
    if family in ("lumina7b", "anole7b"):          # Anole-7B is the Chameleon-7B architecture; config 5 runs it on an fp8 KV cache
        margs, dt, window = BB.LUMINA_7B, torch.bfloat16, 16
    else:
        margs, dt, window = BB.EMU3_8B, torch.float16, 32
    with torch.device(dev):
        model = BB.ChameleonBackbone(margs, attn=ops.HipWindowAttention()).to(dt).eval()
    if family == "emu3_8b":
        model.G1_CFG = dict(model.G1_CFG_EMU3)
    synthetic.fill_state_dict_device(model, seed=0, embed_token_scale=0.7)
    model.enable_fused(ops, gemm="sjd")
    V = margs.vocab_size
    if family == "lumina7b":
        grid = 48
        prompt = lumina_prompt(prompt_len, grid, grid, seed=seed)
        spec = lumina_window_spec(prompt, dev)
        grammar, rules_fn, no_cfg = LuminaGrammar(2000, 10), (lambda c, n: O.lumina_rules(c, n, 2000, 10)), O.lumina_force_no_cfg
        cfg = SJDConfig(jacobi_loop_interval_l=0, jacobi_loop_interval_r=grid * grid + grid - 13, max_num_new_tokens=window, guidance_scale=3.0,
                        seed=seed, max_length=len(prompt) + new_tokens, eos_token_ids=(8196,))
    elif family == "anole7b":
        from sjd_amd.grammar import AnoleGrammar
        n_img = min(1024, max(new_tokens - 1, window + 8))           # 512x512 -> 1024 image tokens + <eoi>, no line tokens
        prompt = synthetic.synthetic_prompt(prompt_len - 1, seed, lo=9000, hi=60000)[0].tolist() + [8197]
        spec = lumina_window_spec(prompt, dev)
        max_len = len(prompt) + n_img + 1
        grammar, no_cfg = AnoleGrammar(V, len(prompt), max_len, n_img), None
        rules_fn = lambda c, n: O.anole_rules(c, n, V, len(prompt), max_len, n_img, top_k=2000)
        cfg = SJDConfig(jacobi_loop_interval_l=0, jacobi_loop_interval_r=n_img - window - 2, max_num_new_tokens=window, guidance_scale=3.0,
                        seed=seed, max_length=max_len, eos_token_ids=(8196,))
    else:
        tok = dict(img_token=151851, eoi_token=151853, eos_token=151850, eol_token=151846, eof_token=151847, pad_token=151643)
        Hh = Ww = 90
        pos = synthetic.synthetic_prompt(prompt_len - 1, seed, lo=1000, hi=150000)[0].tolist() + [tok["img_token"]]
        neg = synthetic.synthetic_prompt(11, seed + 1, lo=1000, hi=150000)[0].tolist() + [tok["img_token"]]
        spec = emu3_window_spec(pos, neg, tok["pad_token"], dev)
        prompt = spec.first_tokens[0].tolist()
        grammar = Emu3Grammar(Hh, Ww, 151854, 32768, top_k=2048, **tok)
        rules_fn, no_cfg = (lambda c, n: O.emu3_rules(c, n, Hh, Ww, 151854, 32768, top_k=2048, **tok)), None
        cfg = SJDConfig(jacobi_loop_interval_l=1, jacobi_loop_interval_r=Hh * Ww - 1, max_num_new_tokens=window, guidance_scale=3.0,
                        seed=seed, max_length=len(prompt) + new_tokens, eos_token_ids=(tok["eos_token"],))
    model.setup_cache(batch=2, s_max=((len(prompt) + new_tokens + 2 * window + 64 + 31) // 32) * 32, dtype=ops.FP8 if family == "anole7b" else None)
    eng = SJDEngine(model, V, dev, max_window=window, use_graph=True)
    O.set_threads(8)
    if stream:       # long decodes (a whole image): the oracle consumes every iteration while the engine produces the next
        rec = _StreamRecorder()
        eng.hook = rec
        (seq, stats), (seq_ref, tr, checks) = _replay_while_decoding(rec, lambda: eng.decode(prompt, spec, grammar, cfg), prompt, rules_fn,
                                                                      _loop_cfg(cfg), V, no_cfg_fn=no_cfg, device=device)
    else:
        rec = _Recorder()
        eng.hook = rec
        seq, stats = eng.decode(prompt, spec, grammar, cfg)
        seq_ref, tr, checks = _replay(rec, prompt, rules_fn, _loop_cfg(cfg), V, no_cfg_fn=no_cfg, device=device)
    assert seq == seq_ref, "token sequences differ"
    assert stats.matched == tr.matched and stats.nfe == len(tr.matched)
    graphs = eng.captured_column_windows()
    return dict(tokens=len(seq) - len(prompt), nfe=stats.nfe, accepted=sorted(set(stats.matched[1:])), n_split=model.attn.n_split,
                fwd_graphs=len(graphs), head_cols=graphs)


@torch.no_grad()
def teacher_forced_real_shape_batch_check(device="cuda:0", n_prompts=4, prompt_lens=(652, 700, 671, 689), new_tokens=56, seed=21):
    """SJDBatchEngine at the REAL Lumina-mGPT-7B shapes and its production launch configuration: 128 window rows per forward on the
    sub-tiled G1 (G1_CFG_128ROW / G1_CFG_64ROW for two prompts), K1 over 2 x n_prompts batch rows (single key split: direct output),
    hipGraph; prompts of different length (different kv_len per slot; max_length is per engine, so the shorter prompts decode up to 48
    tokens more), 25..50 iterations, every slot replayed against the oracle."""
    import sjd_amd.ops as ops
    import sjd_amd.backbones as BB
    import sjd_amd.synthetic as synthetic
    from sjd_amd.engine import SJDConfig
    from sjd_amd.engine_batch import SJDBatchEngine
    from sjd_amd.frontends import lumina_window_spec, lumina_prompt
    from sjd_amd.grammar import LuminaGrammar
    dev = torch.device(device)
    margs, window, grid = BB.LUMINA_7B, 16, 48
    with torch.device(dev):
        model = BB.ChameleonBackbone(margs, attn=ops.HipWindowAttention()).to(torch.bfloat16).eval()
    model.G1_CFG = dict(model.G1_CFG_64ROW if n_prompts == 2 else model.G1_CFG_128ROW)
    synthetic.fill_state_dict_device(model, seed=0, embed_token_scale=0.7)
    model.enable_fused(ops, gemm="sjd")
    V = margs.vocab_size
    prompts = [lumina_prompt(prompt_lens[i % len(prompt_lens)], grid, grid, seed=seed + i) for i in range(n_prompts)]
    specs = [lumina_window_spec(p_, dev) for p_ in prompts]
    max_len = max(len(p_) for p_ in prompts) + new_tokens
    model.setup_cache(batch=2 * n_prompts, s_max=((max_len + 2 * window + 64 + 31) // 32) * 32)
    # every prompt stops after new_tokens tokens of ITS OWN sequence: max_length is per engine, so the longest prompt decides and the
    # shorter ones decode a little further -- each replay below uses the same limit
    cfg = SJDConfig(jacobi_loop_interval_l=0, jacobi_loop_interval_r=grid * grid + grid - 13, max_num_new_tokens=window, guidance_scale=3.0,
                    seed=seed, max_length=max_len, eos_token_ids=(8196,))
    eng = SJDBatchEngine(model, V, dev, n_prompts, max_window=window, use_graph=True)
    recs = [_Recorder() for _ in range(n_prompts)]
    eng.hook = lambda i, d: recs[i](d)
    results = eng.decode_many(prompts, specs, [LuminaGrammar(2000, 10) for _ in range(n_prompts)], cfg)
    O.set_threads(8)
    out = []
    for i, (seq, stats) in enumerate(results):
        c = _loop_cfg(cfg)
        c.seed = cfg.seed + i
        seq_ref, tr, _ = _replay(recs[i], prompts[i], lambda cx, n: O.lumina_rules(cx, n, 2000, 10), c, V, no_cfg_fn=O.lumina_force_no_cfg,
                                 device=device)
        assert seq == seq_ref, f"slot {i}: token sequences differ"
        assert stats.matched == tr.matched, f"slot {i}: accept lengths differ"
        out.append(dict(tokens=len(seq) - len(prompts[i]), nfe=stats.nfe, max_accept=max(stats.matched[1:])))
        recs[i].items.clear()
    return dict(slots=out, n_split=model.attn.n_split, fwd_graphs=len(eng.captured_column_windows()))
