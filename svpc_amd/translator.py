"""Greedy caption decoding on the MI355X — drop-in for the reference's ``Translator`` (src/translator.py:27-228).

Same constructor and ``translate_batch`` contract (12-element ``model_inputs`` list in, ``(dec_seq_list, oov_word_dict)``
out, one ``(S_b, Lt)`` int64 id matrix per video, extended ids ≥ V kept for copied OOV words).  Restructured for the GPU:
all videos of the batch are decoded together — clip encoder, step-wise encoder and visual simulator run once over every
clip of the batch, then the ``max_t_len`` decoding iterations advance all T = Σ_b S_b sentences at once with arg-max, UNK
suppression and OOV→UNK remapping done by a kernel (no host synchronisation inside the loop).

Decoding is **incremental** by default: iteration i embeds only the token chosen at i-1, appends its self-attention K|V row to a
per-layer cache and attends to the i+1 cached keys; cross-attention K|V of the memory slots and the pointer's bank projection
are computed once.  Under the reference's causal∧pad mask (translator.py:88-100 re-runs all Lt positions every iteration) position
i's output depends on exactly those keys, so the ids are the same — ``incremental=False`` keeps the re-run-everything form, and
the parity tests check both against the reference's ids.
"""
from __future__ import annotations

import torch

from . import ops
from .model import BatchPlan, _Ctx
from .ops_common import ACT_RELU, Idx
from .synthetic import BOS, PAD, UNK


class Translator(object):
    def __init__(self, opt, checkpoint, model=None, incremental=True, graph=False):
        self.opt = opt
        self.incremental = incremental
        self.graph = graph          # replay the decode of a recurring batch structure as one hipGraph (see _decode_graphed)
        # graph mode, ≥ 2·min_half videos: the two halves of the batch as two graphs on two streams.  OFF: measured once on the MI355X
        # (round 5, 64 videos × 12 clips, bf16x3) the two replays did not overlap — 36.3 ms per batch against 27.3 ms for the one graph
        # (encoder side 13.3 ms + 2 × 22 iterations of ≈0.5 ms: every half's iterations cost what the whole batch's cost, and the
        # second half's ran after the first's) — DESIGN §0.  Kept as an option because the ids are identical either way (tested).
        self.two_streams = False
        self.min_half = 8
        self.phase_events = None    # bench.py: a list → _decode_core (eager) appends three HIP events: start, encoder side done, end
        self._preps = {}
        self._ttab = None
        self.device = torch.device("cuda" if getattr(opt, "cuda", True) else "cpu")
        self.model_config = checkpoint["model_cfg"]
        self.max_t_len = self.model_config.max_t_len
        self.max_v_len = self.model_config.max_v_len
        self.num_hidden_layers = self.model_config.num_hidden_layers
        model.load_state_dict(checkpoint["model"])
        self.model = model
        self.model.eval()
        if next(model.parameters()).is_cuda:
            # inference without an optimizer: adopt the parameters in a WeightStore, so the packed [Wq Wk Wv] / stacked memory K|V blocks
            # and the bf16 (+ lo plane) weight shadows are resident — without one every packed projection concatenates its weights per call
            from .optim import WeightStore
            WeightStore.for_model(model)

    @classmethod
    def prepare_video_only_inputs(cls, input_ids, input_masks, segment_ids):
        """reference: translator.py:205-228 — overwrites the caller's text half in place, as the reference does."""
        if isinstance(input_ids, list):
            for e1, e2, e3 in zip(input_ids, input_masks, segment_ids):
                text_mask = e3 == 1
                e1[text_mask] = PAD
                e2[text_mask] = 0
            return input_ids, input_masks
        text_mask = segment_ids == 1
        input_ids[text_mask] = PAD
        input_masks[text_mask] = 0
        return input_ids, input_masks

    def translate_batch(self, model_inputs, use_beam=False, recurrent=True, untied=False, xl=False, mtrans=False):
        (input_ids_list, video_features_list, input_masks_list, token_type_ids_list, ingr_input_ids, ingr_masks,
         ingr_sep_masks, ingr_id_dict, oov_word_dict, alignments, actions, batch_step_num) = model_inputs
        return self.translate_batch_greedy(input_ids_list, video_features_list, input_masks_list, token_type_ids_list,
                                           ingr_input_ids, ingr_masks, ingr_sep_masks, ingr_id_dict, oov_word_dict,
                                           alignments, actions, batch_step_num, self.model)

    # ------------------------------------------------------------------ host part: everything that depends on the batch STRUCTURE only
    def _prepare(self, model, batch_step_num, ingr_sep_masks, ingr_id_dict, oov_word_dict, S_pad, N, L, dev):
        cfg = model.config
        mode = cfg.model_mode
        Lt, V = cfg.max_t_len, cfg.vocab_size
        sep_t = torch.as_tensor(ingr_sep_masks).cpu()
        dicts = ingr_id_dict if mode != "video" else [{}] * N
        n_oov = [len(d) if mode != "video" else 0 for d in oov_word_dict]
        key = (tuple(int(v) for v in batch_step_num), sep_t.numpy().tobytes(), tuple(n_oov), S_pad, N, L, str(dev), self.incremental,
               tuple(tuple((int(e), tuple(int(i) for i in lst)) for e, lst in d.items()) for d in dicts))
        prep = self._preps.get(key)
        if prep is not None:
            return prep
        spans = model.ingredient_embeddings.spans(sep_t)   # one host read per batch structure
        plan = model.plan_for(batch_step_num, spans[3], S_pad, N, L, dev)
        T = plan.T
        c_list = [V + x_ for x_ in n_oov]
        prep = dict(key=key, spans=spans, plan=plan, c_list=c_list, T=T,
                    pl=model._ptr_plan(dicts, c_list, Lt, plan.step_ne, plan.row_vid),
                    row_x=Idx([n_oov[b] for b in plan.row_vid.host]),
                    pl1=model._ptr_plan(dicts, c_list, 1, plan.step_ne, plan.step_vid),
                    row_x1=Idx([n_oov[b] for b in plan.step_vid.host]), seq_cross={}, graph=None)
        if len(self._preps) > 32:
            self._preps.clear()
        self._preps[key] = prep
        return prep

    def _text_table(self, model, Lt, cx, dev):
        """(Lt, V, D): text_embeddings(token v at position p) for every (p, v) — inference weights are constant, so the embedding stack
        (word vector → LayerNorm → projection → ReLU → LayerNorm, + position encoding; model.py:484-513) is evaluated once per checkpoint
        for the whole vocabulary with the product's own kernels and the decoding iterations gather rows.  None outside eval mode."""
        if model.training or dev.type != "cuda":
            return None
        te = model.text_embeddings
        ps = [te.word_embeddings.weight] + list(te.word_fc.parameters())
        from .optim import WEIGHTS_EPOCH
        key = (tuple((p.data_ptr(), p._version) for p in ps), WEIGHTS_EPOCH[0], ops.get_precision(), Lt)
        if self._ttab is None or self._ttab[0] != key:
            V = te.word_embeddings.weight.shape[0]
            ids = torch.arange(V, dtype=torch.int32, device=dev)
            base = te.word_fc.run(te.word_embeddings.weight, cx.eps, src_rows=ids, pad_row=PAD)
            self._ttab = (key, (base.unsqueeze(0) + te.position_embeddings_text.pe[:Lt].unsqueeze(1)).contiguous())
        return self._ttab[1]

    # ------------------------------------------------------------------ device part: encoder side once, then the decoding iterations
    def _decode_core(self, model, prep, feats, ids_all, masks_all, ingr_ids_flat):
        cfg = model.config
        mode = cfg.model_mode
        plan, T = prep["plan"], prep["T"]
        dev = feats.device
        Lt, D = cfg.max_t_len, cfg.hidden_size
        cx = _Ctx(cfg, False, model.rng(dev))
        stamps = self.phase_events if (self.phase_events is not None and not torch.cuda.is_current_stream_capturing()) else None

        def stamp():
            if stamps is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                stamps.append(e)
        stamp()
        ents = model.ingredient_embeddings.run(ingr_ids_flat, prep["spans"], cx)
        cls = model._encode_clips(feats, plan.video_rows, ops.take_rows(ids_all, plan.video_rows),
                                  ops.take_rows(masks_all, plan.video_rows), plan.seq_enc, cx,
                                  cls_only=(plan.cls_rows_dev, plan.seq_enc_cls))
        x = ops.span_mean(cls, plan.arange_T, plan.ones_T, add=model.step_positional_encoding.pe, add_idx=plan.step_idx)
        g = model.step_wise_encoder.run(x, plan.seq_step, None, cx)
        if mode in ("full", "reason_copy"):
            _, _, ebar, eall, fbar = model.reasoner.run(g, ents, plan.sim, cx)
            went = ops.linear(ebar, model.Went[0].weight, model.Went[0].bias, act=ACT_RELU)
            wac = ops.linear(fbar, model.Wac[0].weight, model.Wac[0].bias, act=ACT_RELU)
            mem = torch.stack([g, went, wac], 1).reshape(T * 3, D)
            bank = eall
        elif mode == "copy":
            mean_ing = ops.span_mean(ents, plan.ent_off, plan.ent_len)
            mem = torch.stack([g, ops.take_rows(mean_ing, plan.step_vid_dev)], 1).reshape(T * 2, D)
            bank = model._padded_bank(ents, plan)
        else:
            mem = g
            bank = None

        stamp()          # encoder side (clip encoder, step encoder, simulator, memory) done; the Lt decoding iterations follow
        text = torch.full((T, Lt), PAD, dtype=torch.int32, device=dev)
        ext = torch.full((T, Lt), PAD, dtype=torch.int32, device=dev)
        nxt = torch.full((T,), BOS, dtype=torch.int32, device=dev)
        nxt_ext = nxt.clone()
        if self.incremental:
            n_mem = mem.shape[0] // T
            layers = model.decoder.layer
            caches = [torch.zeros(T * Lt, 2 * D, dtype=torch.float32, device=dev) for _ in layers]
            st = model.decoder.stacked_memory_kv()
            if st is not None:                 # one projection of the memory rows for the whole stack; each layer reads its columns
                mem_kv = ops.split_cols(ops.linear(mem, st[0], st[1]), len(layers))
            else:
                mem_kv = [layer.memory_kv(mem) for layer in layers]
            proj = model.bank_projection(bank) if bank is not None else None
            seq_cross = prep["seq_cross"].get(n_mem)
            if seq_cross is None:
                seq_cross = prep["seq_cross"][n_mem] = ops.SeqInfo(list(range(T)), [1] * T, [s_ * n_mem for s_ in range(T)],
                                                                   [n_mem] * T, dev)
            pl1, row_x1 = prep["pl1"], prep["row_x1"]
            tab = self._text_table(model, Lt, cx, dev)
            text[:, 0] = nxt                    # (position i + 1 of both id matrices is written by iteration i's pick)
            ext[:, 0] = nxt_ext
            for i in range(Lt):
                seq_self = plan.seq_dec_incremental(i, Lt, dev)
                # the text embedding of (token, position) from the per-checkpoint table: one row gather instead of
                # LayerNorm(300) → projection → LayerNorm + position encoding per iteration
                x = ops.take_rows(tab[i], nxt) if tab is not None else model.text_embeddings.run_at(nxt, i, cx)
                for layer, cache, kv in zip(layers, caches, mem_kv):
                    x = layer.step(x, i, Lt, cache, kv, seq_self, seq_cross, cx)
                if mode == "video":
                    scores = model.decoder_classifier.run(x, cx.eps)          # raw logits (translator.py:159)
                else:
                    scores, _ = model._lm_probs(x, bank, pl1, cx, proj=proj)
                nxt_ext, nxt = ops.greedy_pick(scores, pl1["row_c"], row_x1, 1, 0, UNK, append=(text, ext, i + 1) if i + 1 < Lt else None)
        else:
            pl, row_x = prep["pl"], prep["row_x"]
            tmask = torch.zeros(T, Lt, dtype=torch.float32, device=dev)
            for i in range(Lt):
                text[:, i] = nxt
                ext[:, i] = nxt_ext
                tmask[:, i] = 1.0
                xt = model.text_embeddings.run(text.reshape(-1), Lt, cx)
                dec = model.decoder.run(xt, tmask.reshape(-1), mem, plan.seq_dec_self, plan.seq_dec_cross, None, cx)
                if mode == "video":
                    scores = model.decoder_classifier.run(dec, cx.eps)          # raw logits (translator.py:159)
                else:
                    scores, _ = model._lm_probs(dec, bank, pl, cx)
                nxt_ext, nxt = ops.greedy_pick(scores, pl["row_c"], row_x, Lt, i, UNK)
        stamp()
        return text if mode == "video" else ext

    @torch.no_grad()
    def translate_batch_greedy(self, input_ids_list, video_features_list, input_masks_list, token_type_ids_list,
                               ingr_input_ids, ingr_masks, ingr_sep_masks, ingr_id_dict, oov_word_dict, alignments, actions,
                               batch_step_num, rt_model):
        model = rt_model
        dev = video_features_list[0].device
        # the text half of every step's ids / masks is blanked in place, as the reference does (translator.py:205-228).  When the per-step
        # tensors are consecutive slices of one buffer (the usual collate output; model._stacked sees it) that is three launches on the
        # stacked views instead of three per step
        stk = None
        if isinstance(input_ids_list, list) and dev.type == "cuda":
            views = [model._stacked(l) for l in (input_ids_list, input_masks_list, token_type_ids_list)]
            if all(v.data_ptr() == l[0].data_ptr() for v, l in zip(views, (input_ids_list, input_masks_list, token_type_ids_list))):
                stk = views
        if stk is not None:
            self.prepare_video_only_inputs(*stk)
        else:
            input_ids_list, input_masks_list = self.prepare_video_only_inputs(input_ids_list, input_masks_list, token_type_ids_list)
        N, L, F = video_features_list[0].shape
        S_pad = len(input_ids_list)
        feats4 = model._stacked(video_features_list)                                             # (S, N, L, F): zero-copy for a stacked loader
        ids3 = (stk[0] if stk is not None else torch.stack(input_ids_list))                      # (S, N, L)
        masks3 = (stk[1] if stk is not None else torch.stack(input_masks_list))
        ingr_all = torch.as_tensor(ingr_input_ids).to(dev).to(torch.int32)                       # (N, Li)
        sep_all = torch.as_tensor(ingr_sep_masks)

        def part(lo, hi, stream=None):
            n = hi - lo
            prep = self._prepare(model, list(batch_step_num[lo:hi]), sep_all[lo:hi], list(ingr_id_dict[lo:hi]), list(oov_word_dict[lo:hi]),
                                 S_pad, n, L, dev)
            src = (feats4[:, lo:hi], ids3[:, lo:hi], masks3[:, lo:hi], ingr_all[lo:hi])
            if not self.graph or dev.type != "cuda":
                out = self._decode_core(model, prep, src[0].reshape(S_pad * n * L, F), src[1].reshape(-1).to(torch.int32),
                                        src[2].reshape(-1).float(), src[3].reshape(-1))
            else:
                out = self._decode_graphed(model, prep, src, (S_pad, n, L, F), stream)
            return prep["plan"], out

        # Two halves on two streams: the encoder side of a batch is bound by the matrix cores, the Lt decoding
        # iterations are chains of small dependent launches that leave the chip idle — two independent chains interleave.  Videos are
        # independent (the reference decodes them one by one, translator.py:175-191), so the ids are those of the unsplit decode bit for
        # bit (tests/test_config5_gpu.py).  Two replayed graphs on two streams, not one forked graph.
        halves = [(0, N)]
        if self.graph and self.two_streams and dev.type == "cuda" and N >= 2 * self.min_half:
            halves = [(0, N // 2), (N // 2, N)]
        if len(halves) == 1:
            plans_outs = [part(0, N)]
        else:
            from .graph import ops_stream, ops_stream_b
            cur = torch.cuda.current_stream()
            sa = cur if cur != torch.cuda.default_stream() else ops_stream()
            sb = ops_stream_b()
            plans_outs = []
            for (lo, hi), st in zip(halves, (sa, sb)):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    plans_outs.append(part(lo, hi, st))
            for st in (sa, sb):
                cur.wait_stream(st)
        res = []
        for plan, out in plans_outs:
            out = out.to(torch.int64)                  # (one cast per part: the per-video results are views of it)
            for b in range(plan.N):
                o, n_ = plan.h_step_off[b], plan.h_step_len[b]
                res.append(out[o:o + n_])
        return res, oov_word_dict

    def _decode_graphed(self, model, prep, src, shape, stream=None):
        """The ≈2,000 launches of a batch's decode (encoder side + Lt iterations × 6 layers) replayed as ONE hipGraph.  The graph
        is tied to the batch structure (step / ingredient / OOV counts: ``_prepare``'s key): the first batch of a structure runs
        eagerly twice (warm-up: index tables reach the device, kernels set their attributes) and is then captured; later batches of
        the same structure copy their tensors into the captured inputs and replay.  ``src``: (features (S, n, L, F), ids (S, n, L),
        masks (S, n, L), ingredient ids (n, Li)) — possibly strided views of the whole batch (the copy into the captured inputs gathers them)."""
        # a captured graph holds raw pointers: it is valid only while the parameters (and, on the bf16 path, their shadow) live where
        # they lived at capture time — a weight store built later (optimizer start, WeightStore.for_model) re-points them
        from . import ops as _ops
        from .optim import WEIGHTS_EPOCH
        # (epoch + the embedding parameters' version counters: the graph holds the address of the per-checkpoint text-embedding table,
        # which ``_text_table`` rebuilds — elsewhere — when either moves: load_state_dict / copy_ into the same tensors bump only the versions)
        te = model.text_embeddings
        S_pad, n, L, F = shape
        from .graph import ops_stream
        cur = torch.cuda.current_stream()
        side = stream if stream is not None else (cur if cur != torch.cuda.default_stream() else ops_stream())
        sig = (tuple(p.data_ptr() for p in list(model.parameters())[:8]), _ops.get_precision(),
               id(getattr(model, "_svpc_weight_store", None)), WEIGHTS_EPOCH[0],
               tuple(p._version for p in [te.word_embeddings.weight] + list(te.word_fc.parameters())), side.cuda_stream)
        if not isinstance(prep["graph"], dict):
            prep["graph"] = {}                       # one graph per replay stream: two halves of ONE structure must not share captured buffers
        gkey = side.cuda_stream
        g = prep["graph"].get(gkey)
        if g is not None and g[3] != sig:
            g = None
            prep["graph"].pop(gkey, None)

        def flat(st):
            return (st[0].view(S_pad * n * L, F), st[1].view(-1), st[2].view(-1), st[3].view(-1))
        if g is None:
            static = [torch.empty(S_pad, n, L, F, dtype=torch.float32, device=src[0].device),
                      torch.empty(S_pad, n, L, dtype=torch.int32, device=src[0].device),
                      torch.empty(S_pad, n, L, dtype=torch.float32, device=src[0].device),
                      torch.empty(src[3].shape, dtype=torch.int32, device=src[0].device)]
            for dst, s_ in zip(static, src):
                dst.copy_(s_)
            # warm-up and capture on ONE stream per half (svpc_amd.graph.ops_stream / ops_stream_b): a fresh stream per batch
            # structure would pin another 256 MB kernel workspace each (ops._ws is per (device, stream))
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._decode_core(model, prep, *flat(static))
                self._decode_core(model, prep, *flat(static))
            cur.wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            from .graph import capturing
            with capturing(graph, stream=side):
                out = self._decode_core(model, prep, *flat(static))
            g = prep["graph"][gkey] = (graph, static, out, sig)
        graph, static, out, _ = g
        for dst, s_ in zip(static, src):
            dst.copy_(s_)
        graph.replay()
        return out.clone()
