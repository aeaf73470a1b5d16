"""Token routing: the reference's masks compiled ONCE per batch into the index data the HIP
kernels consume (``moka_routing`` in include/moka_hip.h).

The reference re-derives indices from the masks inside every one of the 224 adapter calls
of a forward, with ~10 host syncs per call (VT ``layer.py:603-667``: ``.any()``,
``.nonzero()``, ``torch.where``; AVT ``lora.py:489,512``: ``torch.where`` per sample).  Here
the masks are turned into ``tok_mod`` / key positions / ``klen`` on the device with a single
host read-back per batch (needed for the key-block size and to raise the reference's
errors), and the result is cached on the identity of the mask tensors.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import List, Optional, Sequence

import torch

from . import _lib

MOD_NONE = _lib.MOKA_MOD_NONE


class MokaRouting:
    """Device-resident routing of one batch.  ``struct`` is the C view passed to the library."""

    def __init__(self, tok_mod: torch.Tensor, kpos: torch.Tensor, klen: torch.Tensor,
                 B: int, S: int, Lk_max: int, M: int):
        dev = tok_mod.device
        Lkp = max(Lk_max, 1)
        if kpos.shape[1] != Lkp:
            kpos = torch.full((B, Lkp), -1, dtype=torch.int32, device=dev)
        self.tok_mod, self.kpos, self.klen = tok_mod, kpos.contiguous(), klen.contiguous()
        self.B, self.S, self.Lk_max, self.M = B, S, Lk_max, M
        self.T = B * S
        # flat key token index per slot (-1: zero row -- padding slot, invalid key, or a key token of no modality)
        live = (self.kpos >= 0) & (torch.arange(Lkp, device=dev)[None, :] < self.klen[:, None])
        flat = (torch.arange(B, device=dev)[:, None] * S + self.kpos.clamp(min=0)).long()
        live = live & (tok_mod[:B * S][flat.clamp(max=B * S - 1)] != MOD_NONE)
        self.ktok = torch.where(live, flat, torch.full_like(flat, -1)).to(torch.int32).contiguous()
        # inverse map token -> key slot (a key row is finished by the second half of moka_cross_bwd)
        kslot = torch.full((B * S,), -1, dtype=torch.int32, device=dev)
        if Lk_max > 0:
            slots = torch.arange(Lkp, device=dev, dtype=torch.int32)[None, :].expand(B, Lkp)
            kslot[flat[live]] = slots[live]
        self.kslot = kslot.contiguous()
        self._ws = {}
        self.struct = _lib.MokaRoutingStruct(self.tok_mod.data_ptr(), self.ktok.data_ptr(), self.klen.data_ptr(),
                                             self.kslot.data_ptr(), B, S, Lk_max, M)
        # tokens in MORE than one modality (from_avt_masks): the routing then describes S = S_real + D tokens per sample, the last D being
        # VIRTUAL tokens -- copies of the rows dup_src [B, D] that carry the token's further memberships (extend / fold below)
        self.dup_src: Optional[torch.Tensor] = None
        self.S_real = S

    # ------------------------------------------------------------------ tokens of several modalities
    def extend(self, x: torch.Tensor) -> torch.Tensor:
        """[B, S_real, d] -> [B, S, d]: the rows of the virtual tokens appended (differentiable: their input gradients flow back to the
        rows they copy)."""
        if self.dup_src is None:
            return x
        x3 = x.reshape(self.B, self.S_real, x.shape[-1])
        idx = self.dup_src[..., None].expand(-1, -1, x3.shape[-1])
        return torch.cat([x3, x3.gather(1, idx)], dim=1)

    def fold(self, y: torch.Tensor) -> torch.Tensor:
        """[B, S, d] -> [B, S_real, d]: every virtual token's row added to the row of the token it stands for (the reference sums the
        modality streams of a token in front of lora_B, ``lora.py:524-530``: the up-projection is linear, so the rows may be summed behind it)."""
        if self.dup_src is None:
            return y
        y3 = y.reshape(self.B, self.S, y.shape[-1])
        idx = self.dup_src[..., None].expand(-1, -1, y3.shape[-1])
        return y3[:, :self.S_real].scatter_add(1, idx, y3[:, self.S_real:])

    def cross_ws(self, r: int, slot: int = 0) -> torch.Tensor:
        """Scratch of moka_cross_bwd for rank r (no initialisation needed; one per routing, rank pad and
        group slot, consumed inside the call, so consecutive layers share it)."""
        rp = _lib.rank_pad(r)
        ws = self._ws.get((rp, slot))
        if ws is None:
            n = int(_lib.load().moka_cross_ws_bytes(self.B, self.S, self.Lk_max, int(r)))
            ws = torch.empty(max(n, 256), dtype=torch.uint8, device=self.tok_mod.device)
            self._ws[(rp, slot)] = ws
        return ws

    @property
    def device(self):
        return self.tok_mod.device

    # ------------------------------------------------------------------ constructors
    @staticmethod
    def _pad_tok_mod(tok_mod_bs: torch.Tensor) -> torch.Tensor:
        T = tok_mod_bs.numel()
        Tpad = (T + 63) // 64 * 64 + 64
        out = torch.full((Tpad,), MOD_NONE, dtype=torch.uint8, device=tok_mod_bs.device)
        out[:T] = tok_mod_bs.reshape(-1)
        return out

    @classmethod
    def from_avt_masks(cls, modality_mask: Sequence[torch.Tensor]) -> "MokaRouting":
        """[text, video, audio, question] masks, {0,1}, shape [B,L,1] or [B,L]
        (AVT ``unified_arch.py:159-240`` builds them, ``lora.py:462-521`` consumes them).
        Keys = contiguous span first..last question token (``lora.py:489-491``), a key row is
        non-zero only where the token is question AND text (``lora.py:482``).
        Raises IndexError when a sample has no question token, as ``lora.py:489-490`` does.

        A token in SEVERAL masks (never in the reference's data, ``unified_arch.py:159-240`` builds disjoint masks; pinned by
        tests/golden/avt_dual_modality.npz): ``lora.py:468-477`` runs every adapter on its masked copy of x, so such a token has one
        rank-space row PER membership, each stream interacts with the question rows on its own (:485-521) and the rows are summed in
        front of ``lora_B0`` (:524-530).  One modality id per token cannot say that, so the routing is built over S + D tokens per
        sample: the token keeps its FIRST membership (text before video before audio -- a question-and-text token stays a text key
        row) and every further membership becomes a virtual token behind the sample's real ones (``dup_src``: the row it copies);
        ``functional.moka_linear`` feeds the kernels the extended rows and folds the virtual outputs / input gradients back."""
        t, v, a, q = [(m.reshape(m.shape[0], m.shape[1]) == 1) for m in modality_mask[:4]]
        B, S = t.shape
        dev = t.device
        tok = torch.full((B, S), MOD_NONE, dtype=torch.uint8, device=dev)
        tok[a] = 2
        tok[v] = 1
        tok[t] = 0                                               # (a token of several modalities keeps its FIRST membership)
        idx = torch.arange(S, device=dev).expand(B, S)
        first = torch.where(q, idx, S).min(dim=1).values
        last = torch.where(q, idx, -1).max(dim=1).values
        klen = (last - first + 1)
        extra = [v & t, a & (t | v)]                             # memberships behind a token's first one (video, audio)
        n_extra = extra[0].sum(dim=1) + extra[1].sum(dim=1)
        stats = torch.stack([(klen <= 0).any().int(), klen.max().int(), n_extra.max().int()]).tolist()   # the one sync
        if stats[0]:
            raise IndexError("index 0 is out of bounds for dimension 0 with size 0")
        if stats[2] > 0:
            D = (int(stats[2]) + 15) // 16 * 16                  # whole 16-token tiles behind every sample
            E = torch.cat(extra, dim=1)                          # [B, 2 S]: slot e = (modality 1 + e // S, position e % S)
            order = torch.sort((~E).to(torch.uint8), dim=1, stable=True).indices
            if D > order.shape[1]:
                order = torch.cat([order, order.new_zeros(B, D - order.shape[1])], dim=1)
            order = order[:, :D]
            live = torch.arange(D, device=dev)[None, :] < n_extra[:, None]
            src = torch.where(live, order % S, torch.zeros_like(order))
            vmod = torch.where(live, 1 + order // S, torch.full_like(order, 255))
            zeros = torch.zeros(B, D, dtype=torch.bool, device=dev)
            ext = [torch.cat([t, zeros], 1), torch.cat([v & ~extra[0], vmod == 1], 1), torch.cat([a & ~extra[1], vmod == 2], 1), torch.cat([q, zeros], 1)]
            rt = cls.from_avt_masks([m.to(torch.int32) for m in ext])          # (disjoint now)
            rt.dup_src, rt.S_real = src.long().contiguous(), S
            return rt
        Lk = int(stats[1])
        kp = first[:, None] + torch.arange(Lk, device=dev)[None, :]
        inside = kp <= last[:, None]
        kpc = kp.clamp(max=S - 1)
        valid = inside & q.gather(1, kpc) & t.gather(1, kpc)
        kpos = torch.where(valid, kp, -1).to(torch.int32).contiguous()
        return cls(cls._pad_tok_mod(tok), kpos, klen.to(torch.int32).contiguous(), B, S, Lk, 3)

    @classmethod
    def from_vt_masks(cls, text_mask: torch.Tensor, image_mask: torch.Tensor,
                      question_mask: torch.Tensor) -> "MokaRouting":
        """bool [B,S] masks (VT ``train.py:206-231``; consumed at ``layer.py:594-669``).
        Keys = exact question index set; a sample without image or question tokens has no
        interaction (``layer.py:630-637``)."""
        B, S = text_mask.shape[0], text_mask.shape[1]
        t = (text_mask.reshape(B, S) == 1)
        i = (image_mask.reshape(B, S) == 1)
        q = (question_mask.reshape(B, S) == 1)
        dev = t.device
        tok = torch.full((B, S), MOD_NONE, dtype=torch.uint8, device=dev)
        tok[t] = 0
        tok[i] = 1
        klen = torch.where(i.any(dim=1), q.sum(dim=1), 0)
        overlap = (t & i).any()
        stats = torch.stack([overlap.int(), klen.max().int()]).tolist()                           # the one sync
        if stats[0]:
            raise ValueError("text and image masks overlap")
        Lk = int(stats[1])
        order = torch.sort((~q).to(torch.uint8), dim=1, stable=True).indices[:, :Lk]
        live = torch.arange(Lk, device=dev)[None, :] < klen[:, None]
        kpos = torch.where(live, order, -1).to(torch.int32).contiguous()
        if Lk == 0:
            kpos = torch.full((B, 1), -1, dtype=torch.int32, device=dev)
        return cls(cls._pad_tok_mod(tok), kpos, klen.to(torch.int32).contiguous(), B, S, Lk, 2)

    # ---- routing straight from what the data pipeline knows, masks never materialised per call
    @classmethod
    def from_vt_batch(cls, input_ids: torch.Tensor, labels: torch.Tensor, image_pad_id: int,
                      attention_mask: Optional[torch.Tensor] = None) -> "MokaRouting":
        """The visual-text recipe (``VisualText/train/train.py:206-231`` per sample, ``:258-318`` right-padding in
        the collator) evaluated for the whole padded batch on the device: image = placeholder id, text = every other
        real token, question = non-image tokens that are not supervised (label -100) and lie after the last image
        token.  ``attention_mask`` marks the real tokens (the collator pads every mask with False); without it every
        position counts as real, as for a batch of one."""
        B, S = input_ids.shape
        real = torch.ones_like(input_ids, dtype=torch.bool) if attention_mask is None else attention_mask.reshape(B, S).bool()
        image = (input_ids == image_pad_id) & real
        text = (input_ids != image_pad_id) & real
        pos = torch.arange(S, device=input_ids.device).expand(B, S)
        last_image = torch.where(image, pos, -1).max(dim=1, keepdim=True).values          # -1: no image token
        after = (pos > last_image) & (last_image >= 0)
        question = text & (labels.reshape(B, S) == -100) & after
        return cls.from_vt_masks(text, image, question)

    @classmethod
    def from_segments(cls, segments: Sequence[Sequence[tuple]], S: int, device, variant: str = "avt",
                      pad: str = "left") -> "MokaRouting":
        """Routing from per-sample segment lists ``[(kind, n_tokens), ...]`` -- what the audio-visual-text embedding
        assembly knows on the host while it concatenates text and feature segments (``AudioVisualText/models/
        unified_arch.py:150-246``: the text run closed by ``<question_end>`` is the question, ``<video>`` / ``<image>`` /
        ``<audio>`` placeholders become feature tokens; ``:306-324`` left-pads the batch).  kinds: ``t`` text, ``q``
        question (text), ``v`` video / image, ``a`` audio, ``p`` explicit padding.  Built with host integers only: no
        mask tensors, no device read-back, one upload.  ``variant='vt'`` maps ``v`` to the image adapter (M = 2) and skips
        samples without image or question tokens, as ``from_vt_masks`` does."""
        import numpy as np
        if variant not in ("avt", "vt"):
            raise ValueError(variant)
        M = 3 if variant == "avt" else 2
        B = len(segments)
        code = {"t": 0, "q": 0, "v": 1, "a": 2, "p": MOD_NONE}
        tok = np.full((B, S), MOD_NONE, dtype=np.uint8)
        qpos: List[List[int]] = []
        for b, segs in enumerate(segments):
            n = sum(int(l) for _, l in segs)
            if n > S:
                raise ValueError(f"sample {b}: {n} tokens do not fit S = {S}")
            at = S - n if pad == "left" else 0
            qs: List[int] = []
            has_image = False
            for kind, l in segs:
                l = int(l)
                if kind not in code or (kind == "a" and variant == "vt"):
                    raise ValueError(f"segment kind {kind!r} is not part of the {variant} layout")
                tok[b, at:at + l] = code[kind]
                if kind == "q":
                    qs.extend(range(at, at + l))
                has_image |= kind == "v" and l > 0
                at += l
            if variant == "avt" and not qs:
                raise IndexError("index 0 is out of bounds for dimension 0 with size 0")     # lora.py:489-490
            if variant == "vt" and not has_image:
                qs = []                                                                        # layer.py:630-637
            qpos.append(qs)
        if variant == "avt":      # contiguous span first..last question token; only question tokens are live key rows
            spans = [list(range(q[0], q[-1] + 1)) for q in qpos]
            live = [set(q) for q in qpos]
        else:
            spans, live = qpos, [set(q) for q in qpos]
        Lk = max((len(sp) for sp in spans), default=0)
        kpos = np.full((B, max(Lk, 1)), -1, dtype=np.int32)
        klen = np.zeros((B,), dtype=np.int32)
        for b, sp in enumerate(spans):
            klen[b] = len(sp)
            for j, p_ in enumerate(sp):
                kpos[b, j] = p_ if p_ in live[b] else -1
        dev = torch.device(device)
        return cls(cls._pad_tok_mod(torch.from_numpy(tok).to(dev)), torch.from_numpy(kpos).to(dev),
                   torch.from_numpy(klen).to(dev), B, S, Lk, M)

    @classmethod
    def plain(cls, B: int, S: int, device, M: int = 1) -> "MokaRouting":
        """Masks None (decode step): every token goes through the text adapter, no interaction
        (AVT ``lora.py:373-381``, VT ``layer.py:672-678``)."""
        tok = torch.zeros((B, S), dtype=torch.uint8, device=device)
        kpos = torch.full((B, 1), -1, dtype=torch.int32, device=device)
        klen = torch.zeros((B,), dtype=torch.int32, device=device)
        return cls(cls._pad_tok_mod(tok), kpos, klen, B, S, 0, M)


def routing_for_samples(B: int, S: int, samples: Sequence[int], device) -> MokaRouting:
    """Plain-LoRA routing of a SUBSET of the batch (per-sample ``adapter_names``, ``layer.py:346-381``): every token of the
    listed samples goes through adapter 0, all other tokens belong to no modality (the kernels skip them), no interaction."""
    tok = torch.full((B, S), MOD_NONE, dtype=torch.uint8, device=device)
    if len(samples):
        tok[torch.as_tensor(list(samples), dtype=torch.long, device=device)] = 0
    kpos = torch.full((B, 1), -1, dtype=torch.int32, device=device)
    klen = torch.zeros((B,), dtype=torch.int32, device=device)
    return MokaRouting(MokaRouting._pad_tok_mod(tok), kpos, klen, B, S, 0, 1)


_OVERRIDE = threading.local()


class use_routing:
    """``with use_routing(rt): model(...)``: every masked adapter call of this thread takes ``rt`` instead of compiling the masks it is
    handed (``RoutingCache.get``).  What a captured training step needs (``schedule.GraphedTrainStep``): compiling masks reads the device
    back once per batch, which a stream capture forbids -- the step's routing lives in ``StaticRouting`` buffers the graph's launches point
    at, refreshed from the batch's masks before every replay."""

    def __init__(self, rt: Optional["MokaRouting"]):
        self.rt = rt

    def __enter__(self):
        self.prev = getattr(_OVERRIDE, "rt", None)
        _OVERRIDE.rt = self.rt
        return self.rt

    def __exit__(self, *exc):
        _OVERRIDE.rt = self.prev
        return False


class StaticRouting(MokaRouting):
    """A routing at FIXED device addresses with room for ``key_capacity`` key slots per sample: the launches of a captured step point at its
    buffers, ``load(rt)`` copies a batch's routing into them (no allocation, no read-back; the source routing was compiled outside the
    capture).  The kernels read ``klen[b]`` keys of sample b, so spare capacity costs nothing but workspace."""

    def __init__(self, like: MokaRouting, key_capacity: Optional[int] = None):
        if like.dup_src is not None:
            raise ValueError("StaticRouting: tokens of several modalities (virtual tokens) change the token count per batch; not capturable")
        dev = like.device
        cap = int(key_capacity) if key_capacity is not None else (like.Lk_max + 63) // 64 * 64 + 64
        cap = max(cap, like.Lk_max, 1)
        self.tok_mod = like.tok_mod.clone()
        self.kpos = torch.full((like.B, cap), -1, dtype=torch.int32, device=dev)
        self.ktok = torch.full((like.B, cap), -1, dtype=torch.int32, device=dev)
        self.klen = like.klen.clone()
        self.kslot = like.kslot.clone()
        self.B, self.S, self.Lk_max, self.M, self.T = like.B, like.S, cap, like.M, like.T
        self._ws = {}
        self.struct = _lib.MokaRoutingStruct(self.tok_mod.data_ptr(), self.ktok.data_ptr(), self.klen.data_ptr(), self.kslot.data_ptr(),
                                             like.B, like.S, cap, like.M)
        self.dup_src, self.S_real = None, like.S
        self.load(like)

    def load(self, rt: MokaRouting) -> None:
        if (rt.B, rt.S, rt.M) != (self.B, self.S, self.M) or rt.dup_src is not None:
            raise ValueError(f"StaticRouting.load: the batch's routing is B={rt.B} S={rt.S} M={rt.M} (virtual tokens: {rt.dup_src is not None}), "
                             f"the captured step was built for B={self.B} S={self.S} M={self.M}")
        if rt.Lk_max > self.Lk_max:
            raise ValueError(f"StaticRouting.load: {rt.Lk_max} key slots exceed the captured capacity {self.Lk_max} (re-capture with a larger key_capacity)")
        n = rt.ktok.shape[1]
        self.tok_mod.copy_(rt.tok_mod)
        self.ktok.fill_(-1)
        self.ktok[:, :n].copy_(rt.ktok)
        self.kpos.fill_(-1)
        self.kpos[:, :n].copy_(rt.kpos)
        self.klen.copy_(rt.klen)
        self.kslot.copy_(rt.kslot)


class RoutingCache:
    """Routing keyed on the identity of the mask tensors (storage pointer, offset, shape, strides, dtype, version
    counter): the decoder passes the very same mask objects to all 7 x n_layers projections of a forward, so one
    host read-back serves the whole batch.

    What the key cannot see is a rewrite that does not bump the version counter (``mask.data[...] = ...``, a raw
    pointer write): such a batch would be routed with the previous batch's masks.  The reference's own pipeline builds
    fresh mask tensors per batch (``unified_arch.py:306-324``, ``train.py:210-231``), which is always safe; for code
    that recycles mask buffers either pass explicit routings (``MokaRouting.from_*``) or set ``MOKA_ROUTING_VERIFY=1``:
    every hit then compares a device-side fingerprint of the masks (one host sync per call -- a debugging mode).
    Tensors without a version counter (created under ``torch.inference_mode()``) are never cached: their routing is
    built on the spot, once per call."""

    def __init__(self, capacity: int = 8):
        self.capacity = capacity
        self._items: List[tuple] = []
        self.verify = os.environ.get("MOKA_ROUTING_VERIFY", "0") not in ("", "0")

    @staticmethod
    def _key(kind: str, masks: Sequence[torch.Tensor]):
        parts = []
        for m in masks:
            try:
                ver = m._version
            except RuntimeError:                 # inference tensors do not track a version counter
                return None
            parts.append((m.data_ptr(), m.storage_offset(), tuple(m.shape), tuple(m.stride()), ver, m.dtype, str(m.device)))
        return (kind,) + tuple(parts)

    @staticmethod
    def _fingerprint(masks: Sequence[torch.Tensor]) -> torch.Tensor:
        """Position-weighted sums of the masks (device tensor, no sync): changes whenever a mask bit moves."""
        out = []
        for m in masks:
            f = m.reshape(-1).to(torch.int64)
            w = torch.arange(1, f.numel() + 1, device=f.device, dtype=torch.int64)
            out.append((f * w).sum())
        return torch.stack(out)

    @staticmethod
    def _build(kind: str, masks: Sequence[torch.Tensor]) -> MokaRouting:
        if kind == "avt":
            return MokaRouting.from_avt_masks(masks)
        if kind == "vt":
            return MokaRouting.from_vt_masks(*masks)
        raise ValueError(kind)

    def clear(self) -> None:
        self._items.clear()

    def get(self, kind: str, masks: Sequence[torch.Tensor]) -> MokaRouting:
        ov = getattr(_OVERRIDE, "rt", None)
        if ov is not None:                       # (use_routing: a captured step's static routing)
            return ov
        key = self._key(kind, masks)
        if key is None:
            return self._build(kind, masks)
        for idx, (k, rt, _keep, fp) in enumerate(self._items):
            if k == key:
                if self.verify and fp is not None and not torch.equal(fp, self._fingerprint(masks)):
                    del self._items[idx]         # rewritten behind the version counter: rebuild
                    break
                return rt
        rt = self._build(kind, masks)
        fp = self._fingerprint(masks) if self.verify else None
        self._items.append((key, rt, list(masks), fp))     # keep the masks alive so data_ptr stays unique
        if len(self._items) > self.capacity:
            self._items.pop(0)
        return rt

    def plain(self, B: int, S: int, device, M: int) -> MokaRouting:
        key = ("plain", B, S, str(device), M)
        for k, rt, _keep, _fp in self._items:
            if k == key:
                return rt
        rt = MokaRouting.plain(B, S, device, M)
        self._items.append((key, rt, None, None))
        if len(self._items) > self.capacity:
            self._items.pop(0)
        return rt


GLOBAL_ROUTING_CACHE = RoutingCache()
