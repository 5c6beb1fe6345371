"""Thin host-side owner of one plipmi handle.  PyTorch-ROCm is used for device
memory, streams and tensors only -- all arithmetic happens inside libplipmi.so."""
from __future__ import annotations

import contextlib
import ctypes as C
import math
from typing import Mapping, Optional

import numpy as np
import torch

from . import _lib
from .config import PlipConfig

_DTYPES = {"fp32": _lib.F32, "f32": _lib.F32, "float32": _lib.F32, torch.float32: _lib.F32,
           "bf16": _lib.BF16, "bfloat16": _lib.BF16, torch.bfloat16: _lib.BF16,
           # IEEE half operands: same matrix-core rate as bf16, 8x smaller operand rounding; the reference's own GPU dtype
           "f16": _lib.F16, "fp16": _lib.F16, "float16": _lib.F16, "half": _lib.F16, torch.float16: _lib.F16}
# bf16 engine: leading text blocks that run on f16 operands by default.  The bf16 engine's cosine error is mostly operand
# rounding in the FIRST text blocks (the residual stream is small there, so a block's rounding error is large against it:
# profiles/r04_text_layer_precision.txt).  Round 4 ran four of them (cosine 8.3e-4 -> 4.7e-4 of the 1e-3 bar, text_embeds 7.0e-4).  Round 5,
# on the final kernels (profiles/r05_text_f16_layers.txt): the dial costs -2.4 / -1.7 / -2.7 / -3.7 % of the step at 4 / 6 / 8 / 12 blocks --
# eight cost what four do -- and eight land at cosine 3.8e-4, text_embeds 4.2e-4 (heavy-tailed checkpoint 2.2e-4 / 2.7e-4): both inside
# VERDICT r4's "cosine <= 4.7e-4 and text_embeds <= 6e-4", which four (7.0e-4) and six (cosine 5.2e-4) are not.  DESIGN.md section 2.1.
# text_f16_layers=0 is the pure bf16 engine, =t_layers the TEXT_TOWER_F16 one.
DEFAULT_TEXT_F16_LAYERS = 8

_TORCH_DTYPE = {_lib.F32: torch.float32, _lib.BF16: torch.bfloat16, _lib.F16: torch.float16}
_SIDE_STREAMS = {}      # device -> the process's first ordinary stream, the first candidate below (created with the first engine)
_PAIR_STREAMS = {}      # (device, main stream) -> the text tower's stream of Engine.encode_pair, CHECKED to overlap with that main stream


def _code(dt) -> int:
    return _DTYPES[dt]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    """One MI355X engine = packed weights + workspace for ``max_batch`` images/captions."""

    def __init__(self, cfg: PlipConfig, state_dict: Mapping[str, object], device="cuda:0", dtype="bf16",
                 max_batch: int = 256, *, ln_fold: bool = True, pooled_last_block: bool = True,
                 pack_captions: bool = False, mfma_attention: bool = True, graph_batch: Optional[int] = None,
                 text_f16: bool = False, text_f16_layers: Optional[int] = None, latency_batch: int = 0,
                 pass_batch: Optional[int] = None, _config_struct_size: Optional[int] = None):
        """``ln_fold`` / ``pooled_last_block`` / ``mfma_attention`` = False select the A/B forms of the 16-bit engines
        (separate LayerNorm kernels, the last block on every token, the exact VALU attention kernel);
        ``text_f16`` (bf16 engine only): the text tower runs on IEEE-half operands, the image tower stays bf16;
        ``text_f16_layers`` (bf16 engine only): only that many LEADING text blocks do (None = the engine default,
        ``DEFAULT_TEXT_F16_LAYERS``; 0 = a pure bf16 engine) -- plipmi_config.text_f16_layers;
        ``graph_batch``: None = default small-batch hipGraph replay (<= 32 samples), 0 = never, n = up to n samples.
        ``pass_batch``: calls of at least twice that many samples run as equal back-to-back passes of at most that many
        (plipmi_config.pass_batch: None = automatic -- 256 for ViT-B/32 --, 0 / negative = never split); same bits either way.
        ``latency_batch``: batches of at most that many samples run on the split-K small-M GEMMs (plipmi_set_latency_batch;
        faster up to batch 8, embeddings then differ from the big-batch path's by up to 6e-4 -- off by default).
        All of it is per-handle configuration (include/plipmi.h plipmi_config.flags): no environment variables.
        (``_config_struct_size``: ABI tests only -- announce a plipmi_config of that many bytes, as a caller compiled against
        an older, shorter header would.)"""
        cfg.validate()
        if not torch.cuda.is_available():
            raise RuntimeError("plip_amd needs a ROCm GPU (MI355X / gfx950): torch.cuda.is_available() is False "
                               "and there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"device must be a cuda(ROCm) device, got {device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.dtype_code = _DTYPES[dtype]
        self.dtype_name = {_lib.BF16: "bf16", _lib.F32: "f32", _lib.F16: "f16"}[self.dtype_code]
        self.flags = ((0 if ln_fold else _lib.FLAG_SEPARATE_LAYERNORM) | (0 if pooled_last_block else _lib.FLAG_DENSE_LAST_BLOCK) |
                      (_lib.FLAG_PACK_CAPTIONS if pack_captions else 0) | (0 if mfma_attention else _lib.FLAG_VALU_ATTENTION) |
                      (_lib.FLAG_TEXT_TOWER_F16 if text_f16 else 0))
        gb = 0 if graph_batch is None else (-1 if int(graph_batch) <= 0 else int(graph_batch))
        if text_f16_layers is None:
            text_f16_layers = min(DEFAULT_TEXT_F16_LAYERS, cfg.t_layers) if (self.dtype_code == _lib.BF16 and not text_f16) else 0
        if text_f16_layers and (self.dtype_code != _lib.BF16 or text_f16):
            raise ValueError("text_f16_layers is a mode of the bf16 engine (and excludes text_f16, which is all of them)")
        self.text_f16_layers = int(text_f16_layers)
        self.max_batch = int(max_batch)
        self.lib = _lib.load()
        self._h = C.c_void_p()
        self.logit_scale = float(np.asarray(state_dict["logit_scale"], dtype=np.float64)) \
            if not torch.is_tensor(state_dict["logit_scale"]) else float(state_dict["logit_scale"])
        with torch.cuda.device(self.device):
            dev = {}
            for k, v in state_dict.items():
                if k == "logit_scale":
                    continue
                t = v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
                dev[k] = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            c = _lib.Config(C.sizeof(_lib.Config), cfg.image_size, cfg.patch_size, cfg.v_width, cfg.v_layers, cfg.v_heads, cfg.v_mlp,
                            cfg.vocab_size, cfg.context_length, cfg.t_width, cfg.t_layers, cfg.t_heads, cfg.t_mlp,
                            cfg.projection_dim, cfg.layer_norm_eps, self.dtype_code, self.max_batch, self.flags, gb,
                            self.text_f16_layers, 0 if pass_batch is None else (int(pass_batch) if int(pass_batch) > 0 else -1))
            if _config_struct_size is not None:
                c.struct_size = int(_config_struct_size)
            w = _lib.Weights()

            def layers(prefix, n):
                arr = (_lib.LayerWeights * n)()
                names = {"ln1": "layer_norm1", "ln2": "layer_norm2", "q": "self_attn.q_proj", "k": "self_attn.k_proj",
                         "v": "self_attn.v_proj", "o": "self_attn.out_proj", "fc1": "mlp.fc1", "fc2": "mlp.fc2"}
                for i in range(n):
                    for short, long in names.items():
                        for suf, key in (("w", "weight"), ("b", "bias")):
                            setattr(arr[i], f"{short}_{suf}", dev[f"{prefix}.encoder.layers.{i}.{long}.{key}"].data_ptr())
                return arr

            vl, tl = layers("vision_model", cfg.v_layers), layers("text_model", cfg.t_layers)
            w.v_class_embedding = dev["vision_model.embeddings.class_embedding"].data_ptr()
            w.v_patch_weight = dev["vision_model.embeddings.patch_embedding.weight"].data_ptr()
            w.v_pos_embedding = dev["vision_model.embeddings.position_embedding.weight"].data_ptr()
            w.v_pre_ln_w = dev["vision_model.pre_layrnorm.weight"].data_ptr()
            w.v_pre_ln_b = dev["vision_model.pre_layrnorm.bias"].data_ptr()
            w.v_post_ln_w = dev["vision_model.post_layernorm.weight"].data_ptr()
            w.v_post_ln_b = dev["vision_model.post_layernorm.bias"].data_ptr()
            w.visual_projection = dev["visual_projection.weight"].data_ptr()
            w.v_layers = vl
            w.t_token_embedding = dev["text_model.embeddings.token_embedding.weight"].data_ptr()
            w.t_pos_embedding = dev["text_model.embeddings.position_embedding.weight"].data_ptr()
            w.t_final_ln_w = dev["text_model.final_layer_norm.weight"].data_ptr()
            w.t_final_ln_b = dev["text_model.final_layer_norm.bias"].data_ptr()
            w.text_projection = dev["text_projection.weight"].data_ptr()
            w.t_layers = tl
            stream = torch.cuda.current_stream(self.device)
            _lib.check(self.lib.plipmi_create(C.byref(c), C.byref(w), C.c_void_p(stream.cuda_stream),
                                              C.byref(self._h)), "plipmi_create")
            stream.synchronize()  # packing done -> the fp32 upload copies can go
            del dev
        self.device_name = self.lib.plipmi_device_name(self._h).decode()
        if self.device not in _SIDE_STREAMS:                                # as early as possible: the first ordinary stream of the process
            _SIDE_STREAMS[self.device] = torch.cuda.Stream(device=self.device)
        self.pass_batch = int(self.lib.plipmi_get_pass_batch(self._h))     # plipmi_config.pass_batch as resolved (0 = never split)
        if latency_batch:
            self.set_latency_batch(latency_batch)

    # ------------------------------------------------------------------
    def clone(self) -> "Engine":
        """A second engine on the SAME packed weights with a workspace of its own (include/plipmi.h plipmi_clone): an engine runs one
        batch per tower at a time, two engines on two streams run consecutive batches of a corpus side by side (``lanes``)."""
        other = object.__new__(Engine)
        other.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_h", "_lanes", "_vis", "pair_stream_ratio")})
        other._h = C.c_void_p()
        other._lanes, other._no_lanes, other.use_lanes = None, 0, False        # a clone is a lane, it does not fan out itself
        with torch.cuda.device(self.device):
            _lib.check(self.lib.plipmi_clone(self._h, C.byref(other._h)), "plipmi_clone")
        return other

    def lanes(self, n: int = 2):
        """``[(engine, stream), ...]`` for a loop over MANY batches of one tower: lane 0 is this engine on the caller's stream, lane 1
        a clone (made on first use, kept) on the stream ``pair_stream`` measured to run beside it.  Give batch k to lane k % n inside
        ``torch.cuda.stream(stream)`` and call ``join_lanes`` before the outputs are used on the caller's stream: the launch boundaries,
        epilogues and the pooled tail of one batch then run under the next batch's GEMMs (configs[3]'s shard: 99 -> 107 k img/s).  Every
        lane waits for the work the caller's stream holds at THIS call (inputs produced there are ready)."""
        main = torch.cuda.current_stream(self.device)
        if n <= 1:
            return [(self, main)]
        if getattr(self, "_lanes", None) is None:
            self._lanes = [self.clone()]
        side = self.pair_stream(main)
        side.wait_stream(main)
        return [(self, main), (self._lanes[0], side)]

    @contextlib.contextmanager
    def lane_loop(self):
        """For host loops over many batches of one tower::

            with eng.lane_loop() as run:
                for batch in batches:
                    outs.append(run(lambda e: e.encode_image_u8(batch)))

        ``run`` gives consecutive calls to alternating lanes (``lanes``) and the block's exit joins them; with ``use_lanes`` off (or
        inside ``encode_pair`` / ``profile``) every call runs on this engine and the caller's stream, as a plain loop would."""
        lanes = self.lanes() if (self.use_lanes and not self._no_lanes) else [(self, None)]
        k = [0]
        main = torch.cuda.current_stream(self.device)

        def run(fn):
            eng, st = lanes[k[0] % len(lanes)]
            k[0] += 1
            if st is None:
                return fn(eng)
            with torch.cuda.stream(st):
                out = fn(eng)
            if torch.is_tensor(out):
                out.record_stream(main)
            return out

        self._no_lanes += 1         # the calls inside are single lanes: no fan-out of a lane's own chunks
        try:
            yield run
        finally:
            self._no_lanes -= 1
            if len(lanes) > 1:
                self.join_lanes(lanes)

    def join_lanes(self, lanes):
        """The caller's stream waits for every lane: outputs of all batches may be used on it afterwards."""
        main = torch.cuda.current_stream(self.device)
        for _, st in lanes[1:]:
            main.wait_stream(st)

    def close(self):
        for other in getattr(self, "_lanes", None) or []:
            other.close()
        self._lanes = None
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.plipmi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chunks(self, n):
        for s in range(0, n, self.max_batch):
            yield s, min(n, s + self.max_batch)

    use_lanes = True        # a call of more than max_batch rows runs its chunks alternately on this engine and on a clone (``lanes``)
    _no_lanes = 0           # > 0 while something else owns the second stream (encode_pair) or counts this handle's launches (profile)

    def _run_chunks(self, n: int, call):
        """``call(engine, a, b)`` for rows a..b of every chunk of at most ``max_batch`` rows.  A call of several chunks is a corpus walked
        through one tower: its chunks go alternately to this engine on the caller's stream and to a clone (same weights, a workspace of
        its own) on the stream measured to run beside it, joined at the end -- the boundaries, epilogues and the pooled tail of one chunk
        run under the next chunk's GEMMs (+8 % on configs[3]'s shard); a row's bits do not depend on the lane."""
        chunks = list(self._chunks(n))
        if len(chunks) < 2 or not self.use_lanes or self._no_lanes:
            for a, b in chunks:
                call(self, a, b)
            return
        lanes = self.lanes()
        for k, (a, b) in enumerate(chunks):
            eng, st = lanes[k % len(lanes)]
            with torch.cuda.stream(st):
                call(eng, a, b)
        self.join_lanes(lanes)

    # ------------------------------------------------------------------
    def encode_image(self, pixels: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """fp32 [B,3,H,W] (any device) -> fp32 [B,P] on the GPU."""
        cfg = self.cfg
        if pixels.dim() != 4 or pixels.shape[1] != 3 or pixels.shape[2] != cfg.image_size or pixels.shape[3] != cfg.image_size:
            raise ValueError(f"Input image size ({tuple(pixels.shape)}) doesn't match model "
                             f"([B,3,{cfg.image_size},{cfg.image_size}]).")
        with torch.cuda.device(self.device):
            px = pixels.to(device=self.device, dtype=torch.float32).contiguous()
            out = torch.empty((px.shape[0], cfg.projection_dim), dtype=torch.float32, device=self.device)
            self._run_chunks(px.shape[0], lambda e, a, b: _lib.check(
                e.lib.plipmi_encode_image(e._h, _ptr(px[a:b]), b - a, _ptr(out[a:b]), int(normalize), e._stream()), "plipmi_encode_image"))
        return out

    def encode_image_u8(self, tiles: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """uint8 [B,H,W,3] RGB tiles already at the model resolution -> fp32 [B,P]; the CLIP
        normalisation is fused into the patch unfold on the GPU."""
        cfg = self.cfg
        if tiles.dtype != torch.uint8 or tiles.dim() != 4 or tuple(tiles.shape[1:]) != (cfg.image_size, cfg.image_size, 3):
            raise ValueError(f"tiles must be uint8 [B,{cfg.image_size},{cfg.image_size},3], got {tiles.dtype} {tuple(tiles.shape)}")
        with torch.cuda.device(self.device):
            t = tiles.to(device=self.device).contiguous()
            out = torch.empty((t.shape[0], cfg.projection_dim), dtype=torch.float32, device=self.device)
            self._run_chunks(t.shape[0], lambda e, a, b: _lib.check(
                e.lib.plipmi_encode_image_u8(e._h, _ptr(t[a:b]), b - a, _ptr(out[a:b]), int(normalize), e._stream()), "plipmi_encode_image_u8"))
        return out

    def encode_text(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                    normalize: bool = False, eos_token_id: Optional[int] = None) -> torch.Tensor:
        """int [B,ctx] token ids -> fp32 [B,P] on the GPU."""
        cfg = self.cfg
        if input_ids.dim() != 2 or input_ids.shape[1] != cfg.context_length:
            raise ValueError(f"input_ids must be [B,{cfg.context_length}], got {tuple(input_ids.shape)} "
                             "(pad/truncate to the context length like the reference, plip.py:58)")
        if input_ids.device.type == "cpu" and input_ids.numel():
            lo, hi = int(input_ids.min()), int(input_ids.max())
            if lo < 0 or hi >= cfg.vocab_size:
                raise IndexError(f"token id out of range [0,{cfg.vocab_size}): min {lo}, max {hi}")
        # device-resident ids are range-checked BY the embedding kernel; the outcome is known once that work has run:
        # check_async() after a synchronisation, or the next encode_text on this engine, raises IndexError
        # (PLIPMI_ERR_TOKEN_ID) -- also from a later chunk of THIS call when B > max_batch: the rows written so far are
        # then not to be used.  Image-side calls never report it.
        eos = cfg.eos_token_id if eos_token_id is None else int(eos_token_id)
        with torch.cuda.device(self.device):
            ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
            mask = None if attention_mask is None else attention_mask.to(device=self.device, dtype=torch.int64).contiguous()
            out = torch.empty((ids.shape[0], cfg.projection_dim), dtype=torch.float32, device=self.device)
            for other in getattr(self, "_lanes", None) or []:       # an id a clone's embedding kernel flagged is this engine's to report
                if self.lib.plipmi_check_async(other._h) != 0:
                    raise IndexError(_lib.last_error())
            self._run_chunks(ids.shape[0], lambda e, a, b: _lib.check(
                e.lib.plipmi_encode_text(e._h, _ptr(ids[a:b]), _ptr(None if mask is None else mask[a:b]), b - a, eos, _ptr(out[a:b]),
                                         int(normalize), e._stream()), "plipmi_encode_text"))
        return out

    def encode_pair(self, pixels: torch.Tensor, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                    normalize: bool = True, overlap: bool = True):
        """Both towers of one step.  With ``overlap`` the text tower is enqueued on a second HIP stream:
        the towers are independent (separate workspaces), so the tail of one tower's GEMM grid -- 150..600
        workgroups over 256 CUs -- is filled by the other tower's kernels instead of idling."""
        n = pixels.shape[0]
        self._no_lanes += 1         # the second stream belongs to the other tower here (four towers at once measured +10 %)
        try:
            return self._encode_pair(pixels, input_ids, attention_mask, normalize, overlap)
        finally:
            self._no_lanes -= 1

    def _encode_pair(self, pixels, input_ids, attention_mask, normalize, overlap):
        cfg = self.cfg          # the image side's shape error comes first, as in CLIPModel.forward, and before anything is enqueued
        ok = (tuple(pixels.shape[1:]) == (cfg.image_size, cfg.image_size, 3)) if pixels.dtype == torch.uint8 else \
            (pixels.dim() == 4 and tuple(pixels.shape[1:]) == (3, cfg.image_size, cfg.image_size))
        if not ok:
            self._encode_image_any(pixels, normalize)       # raises the tower's own ValueError
        n = pixels.shape[0]
        if overlap and self.pass_batch > 0 and n >= 2 * self.pass_batch and input_ids.shape[0] == n:
            # plipmi_config.pass_batch at the level that owns BOTH streams: equal passes, the two towers of a pass joined before the next
            # one starts -- exactly back-to-back pair steps of pass_batch samples (left to the per-tower calls, the faster tower runs a
            # whole pass ahead and a bs = 512 step measured 5 % SLOWER than one pass; profiles/r06_batch_scaling.txt).  Same bits.
            k = -(-n // self.pass_batch)
            bounds = [(i * n // k, (i + 1) * n // k) for i in range(k)]
            parts = [self._encode_pair_two_streams(pixels[a:b], input_ids[a:b], None if attention_mask is None else attention_mask[a:b],
                                                   normalize) for a, b in bounds]
            return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
        if not overlap:
            return self._encode_image_any(pixels, normalize), self.encode_text(input_ids, attention_mask, normalize)
        return self._encode_pair_two_streams(pixels, input_ids, attention_mask, normalize)

    def _encode_image_any(self, pixels: torch.Tensor, normalize: bool):
        """fp32 NCHW pixels or native uint8 [B,H,W,3] tiles (normalisation fused on the GPU) -- the two image inputs of the path"""
        return self.encode_image_u8(pixels, normalize) if pixels.dtype == torch.uint8 else self.encode_image(pixels, normalize)

    def check_async(self, synchronize: bool = True) -> None:
        """Raise IndexError if an earlier ``encode_text`` was given a token id outside the vocabulary -- the reference's
        embedding lookup raises there (plip.py:68; on a GPU at the next synchronisation, like here)."""
        if synchronize:
            torch.cuda.synchronize(self.device)
        for e in [self] + list(getattr(self, "_lanes", None) or []):
            if self.lib.plipmi_check_async(e._h) != 0:
                raise IndexError(_lib.last_error())

    def set_graph_batch(self, max_batch: int):
        """Batches of at most ``max_batch`` samples replay a captured hipGraph (0 = always launch eagerly)."""
        for e in [self] + list(getattr(self, "_lanes", None) or []):
            _lib.check(self.lib.plipmi_set_graph_batch(e._h, int(max_batch)), "plipmi_set_graph_batch")

    def set_latency_batch(self, max_batch: int):
        """Batches of at most ``max_batch`` samples (0 = never, the default) run their GEMMs on the split-K small-M kernel
        (include/plipmi.h plipmi_set_latency_batch): same arithmetic, fp32 summation order of its own."""
        for e in [self] + list(getattr(self, "_lanes", None) or []):
            _lib.check(self.lib.plipmi_set_latency_batch(e._h, int(max_batch)), "plipmi_set_latency_batch")

    def set_text_packing(self, on: bool):
        """Captions packed to their live rows (0 .. EOS): bit-identical text_embeds, cost proportional to the caption
        lengths instead of the padded 77 (include/plipmi.h plipmi_set_text_packing).  Off by default."""
        for e in [self] + list(getattr(self, "_lanes", None) or []):
            _lib.check(self.lib.plipmi_set_text_packing(e._h, int(bool(on))), "plipmi_set_text_packing")

    def streams_overlap(self, a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> float:
        """include/plipmi.h plipmi_streams_overlap: about 1 when kernels of ``a`` and ``b`` run side by side, about 2 when the two
        streams share a hardware queue (one waits for the other).  Synchronises both."""
        r = C.c_float(0.0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.plipmi_streams_overlap(self._h, C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream), C.byref(r)),
                       "plipmi_streams_overlap")
        return float(r.value)

    def pair_stream(self, main: "torch.cuda.Stream") -> "torch.cuda.Stream":
        """The stream ``encode_pair`` runs the text tower on while the vision tower runs on ``main``: the process's first ordinary
        stream if it overlaps with ``main`` (it does in a process that created no streams before the engine -- the bench), otherwise
        the first of a few fresh ones that does; measured once per (device, main stream), about a millisecond each."""
        key = (self.device, main.cuda_stream)
        side = _PAIR_STREAMS.get(key)
        if side is None:
            tried = []
            for k in range(8):
                cand = _SIDE_STREAMS[self.device] if k == 0 and self.device in _SIDE_STREAMS else torch.cuda.Stream(device=self.device)
                ratio = 2.0 if cand.cuda_stream == main.cuda_stream else self.streams_overlap(main, cand)
                tried.append((ratio, k, cand))          # every candidate stays alive until the choice: fresh ones then differ
                if ratio < 1.5:
                    break
            side = _PAIR_STREAMS[key] = min(tried, key=lambda t: t[:2])[2]
            self.pair_stream_ratio = min(tried, key=lambda t: t[:2])[0]
        return side

    def _encode_pair_two_streams(self, pixels, input_ids, attention_mask, normalize):
        main = torch.cuda.current_stream(self.device)
        # ONE side stream per (device, caller's stream) and process, shared by every engine: HIP streams share a handful of hardware
        # queues, and which queue a new stream lands on depends on how many streams the process has created before.  An engine created
        # late (the ninth of a bench run) used to get a side stream on the main stream's queue -- its two towers then ran in order, 4.47
        # instead of 4.28 ms per step, which is what BENCH_r05's "throughput falls with batch" (bs512 54.6 k) mostly was
        # (profiles/r06_batch_scaling.txt).  The stream is picked once by MEASURING that it runs beside the caller's (pair_stream).
        side = self.pair_stream(main)
        side.wait_stream(main)                      # inputs produced on the main stream are ready
        with torch.cuda.stream(side):
            txt = self.encode_text(input_ids, attention_mask, normalize)
        if getattr(self, "pair_vision_priority", False):
            # experiment (tools/gpu_diag.py prio): the longer tower on a high-priority stream of its own, the shorter one
            # filling the gaps -- see profiles/r02_two_stream_priority.txt for what it measured
            if getattr(self, "_vis", None) is None:
                self._vis = torch.cuda.Stream(device=self.device, priority=-1)
            self._vis.wait_stream(main)
            with torch.cuda.stream(self._vis):
                img = self._encode_image_any(pixels, normalize)
            main.wait_stream(self._vis)
            img.record_stream(main)
        else:
            img = self._encode_image_any(pixels, normalize)
        main.wait_stream(side)
        txt.record_stream(main)
        return img, txt

    def l2_normalize_(self, x: torch.Tensor) -> torch.Tensor:
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
        with torch.cuda.device(self.device):
            _lib.check(self.lib.plipmi_l2_normalize(self._h, _ptr(x), x.shape[0], x.shape[1], self._stream()),
                       "plipmi_l2_normalize")
        return x

    def logits(self, image_embeds: torch.Tensor, text_embeds: torch.Tensor, scale: float = 1.0,
               want_text: bool = True, want_argmax: bool = False):
        """scale * image_embeds @ text_embeds.T -> (logits_per_image, logits_per_text|None, argmax|None)."""
        with torch.cuda.device(self.device):
            img = image_embeds.to(device=self.device, dtype=torch.float32).contiguous()
            txt = text_embeds.to(device=self.device, dtype=torch.float32).contiguous()
            ni, nt, d = img.shape[0], txt.shape[0], img.shape[1]
            if txt.shape[1] != d:
                raise ValueError("embedding widths differ")
            lpi = torch.empty((ni, nt), dtype=torch.float32, device=self.device)
            lpt = torch.empty((nt, ni), dtype=torch.float32, device=self.device) if want_text else None
            am = torch.empty((ni,), dtype=torch.int32, device=self.device) if want_argmax else None
            _lib.check(self.lib.plipmi_logits(self._h, _ptr(img), ni, _ptr(txt), nt, d, float(scale), _ptr(lpi), _ptr(lpt),
                                              _ptr(am), self._stream()), "plipmi_logits")
        return lpi, lpt, am

    def topk(self, scores: torch.Tensor, k: int) -> torch.Tensor:
        with torch.cuda.device(self.device):
            sc = scores.to(device=self.device, dtype=torch.float32).contiguous()
            idx = torch.empty((sc.shape[0], k), dtype=torch.int64, device=self.device)
            _lib.check(self.lib.plipmi_topk(self._h, _ptr(sc), sc.shape[0], sc.shape[1], int(k), _ptr(idx),
                                            self._stream()), "plipmi_topk")
        return idx

    def resize_crop_u8(self, images_u8: torch.Tensor, crop: str = "torchvision") -> torch.Tensor:
        """uint8 [B,H,W,3] images of one size -> uint8 [B,n,n,3] tiles at the model resolution: Pillow-exact bicubic
        resize (shortest edge -> n) + centre crop on the GPU; feed the result to :meth:`encode_image_u8`.
        ``crop`` = "torchvision" (OpenAI ``_transform``) or "hf" (``CLIPImageProcessor``), preprocess.crop_offset."""
        from .preprocess import resize_crop_plan
        n = self.cfg.image_size
        if images_u8.dim() != 4 or images_u8.shape[-1] != 3 or images_u8.dtype != torch.uint8:
            raise ValueError("expected uint8 [B,H,W,3]")
        B, H, Wd = int(images_u8.shape[0]), int(images_u8.shape[1]), int(images_u8.shape[2])
        cache = self.__dict__.setdefault("_resize_plans", {})
        if (Wd, H, crop) not in cache:
            plan = resize_crop_plan(Wd, H, n, crop)
            dev = {k: (None if plan[k] is None else torch.from_numpy(plan[k]).to(self.device)) for k in ("xb", "xk", "yb", "yk")}
            if plan["yb"] is not None:
                r0 = int(plan["yb"][:, 0].min())
                r1 = int((plan["yb"][:, 0] + plan["yb"][:, 1]).max())
            else:
                r0, r1 = plan["top"], plan["top"] + n
            cache[(Wd, H, crop)] = (plan, dev, r0, r1 - r0)
        plan, dev, r0, R = cache[(Wd, H, crop)]
        with torch.cuda.device(self.device):
            src = images_u8.to(self.device).contiguous()
            tmp = torch.empty((B, R, n, 3), dtype=torch.uint8, device=self.device)
            dst = torch.empty((B, n, n, 3), dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.plipmi_resize_crop_u8(
                self._h, _ptr(src), B, H, Wd, n, _ptr(dev["xb"]), _ptr(dev["xk"]),
                0 if plan["xk"] is None else plan["xk"].shape[1], plan["left"], _ptr(dev["yb"]), _ptr(dev["yk"]),
                0 if plan["yk"] is None else plan["yk"].shape[1], plan["top"], r0, R, _ptr(tmp), _ptr(dst),
                self._stream()), "plipmi_resize_crop_u8")
        return dst

    def similarity_topk(self, keys: torch.Tensor, space: torch.Tensor, k: int, return_values: bool = False):
        """Top-k rows of ``space`` by dot product for every row of ``keys`` -- ``(keys @ space.T).argsort()[:, -k:][:, ::-1]``
        (plip.py:83-84, retrieval.py:13-16) without the [Nq, Ns] matrix: scores exist only as [<=4096, <=8192] panels."""
        with torch.cuda.device(self.device):
            q = keys.to(device=self.device, dtype=torch.float32).contiguous()
            sp = space.to(device=self.device, dtype=torch.float32).contiguous()
            if q.dim() != 2 or sp.dim() != 2 or q.shape[1] != sp.shape[1]:
                raise ValueError("keys [Nq,D] and space [Ns,D] must share D")
            idx = torch.empty((q.shape[0], int(k)), dtype=torch.int64, device=self.device)
            vals = torch.empty((q.shape[0], int(k)), dtype=torch.float32, device=self.device) if return_values else None
            _lib.check(self.lib.plipmi_similarity_topk(self._h, _ptr(q), q.shape[0], _ptr(sp), sp.shape[0], q.shape[1],
                                                       int(k), _ptr(idx), _ptr(vals) if vals is not None else None,
                                                       self._stream()), "plipmi_similarity_topk")
        return (idx, vals) if return_values else idx

    def hidden(self, tower: str, layer: int, inp: torch.Tensor) -> torch.Tensor:
        """HF ``hidden_states[layer]`` of a tower (parity tests)."""
        cfg = self.cfg
        vision = tower == "vision"
        S, D = (cfg.v_tokens, cfg.v_width) if vision else (cfg.context_length, cfg.t_width)
        with torch.cuda.device(self.device):
            x = inp.to(device=self.device, dtype=torch.float32 if vision else torch.int64).contiguous()
            if x.shape[0] > self.max_batch:
                raise ValueError("batch larger than max_batch")
            out = torch.empty((x.shape[0], S, D), dtype=torch.float32, device=self.device)
            _lib.check(self.lib.plipmi_debug_hidden(self._h, _lib.VISION if vision else _lib.TEXT, int(layer), _ptr(x),
                                                    x.shape[0], _ptr(out), self._stream()), "plipmi_debug_hidden")
        return out

    # ------------------------------------------------------------------
    @contextlib.contextmanager
    def profile(self, result: list):
        """Per-kernel HIP-event timing of everything launched inside the block; rows are appended to ``result``."""
        _lib.check(self.lib.plipmi_profile_enable(self._h, 1), "plipmi_profile_enable")
        self._no_lanes += 1         # the rows are THIS handle's launches: every chunk of a call stays on it
        try:
            yield
        finally:
            self._no_lanes -= 1
            _lib.check(self.lib.plipmi_profile_enable(self._h, 0), "plipmi_profile_enable")
            rows = (_lib.KernelStat * 128)()
            n = C.c_int(0)
            _lib.check(self.lib.plipmi_profile_read(self._h, rows, 128, C.byref(n)), "plipmi_profile_read")
            for r in rows[: n.value]:
                result.append({"name": r.name.decode(), "calls": int(r.calls), "total_ms": float(r.total_ms),
                               "flops": float(r.flops), "bytes": float(r.bytes)})

    @property
    def logit_scale_exp(self) -> float:
        return math.exp(self.logit_scale)


_HEADS = {}


def heads_engine(device="cuda:0") -> Engine:
    """A handle for callers that only need the evaluation heads (l2_normalize / logits / topk / similarity_topk on
    embeddings they already hold -- reproducibility.evaluation): those entry points use no tower weights, so a
    minimal synthetic configuration is enough to own the scratch memory and the stream plumbing."""
    key = str(torch.device(device))
    if key not in _HEADS:
        from . import weights as W
        from .config import get_config
        cfg = get_config("tiny")
        _HEADS[key] = Engine(cfg, W.synthetic_state_dict(cfg, 0), device=device, dtype="f32", max_batch=1)
    return _HEADS[key]
