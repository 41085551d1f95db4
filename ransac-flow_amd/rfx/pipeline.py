"""Batched coarse-to-fine alignment of image pairs on one MI355X.

Restructures the reference's per-pair, host-driven flow (quick_start/align2images.py:53-97 on top of
quick_start/coarseAlignFeatMatch.py:92-173) for throughput: all pairs of a batch go through every
network layer in ONE launch per layer and pyramid scale (batch dimension = pairs), normalised features
are written straight into the concatenated (1024, nA) match matrix, and the host synchronises twice per
batch (match counts -> RANSAC index draw; final results) instead of >= 3 times per 100 RANSAC
hypotheses.  Semantics per pair are the reference's (variant A: ResizeMaxSize, or B: ResizeMinSize).

Host-side work that stays on the CPU: PIL LANCZOS pyramid + ToTensor/Normalize.
``prepare()`` does it and uploads; everything after runs on the device.
"""
import collections
import os

import numpy as np
import torch
import PIL.Image as Image

from . import ops
from .nets import ResNet50Trunk, FeatureExtractorNet, NetFlowCoarseNet, NetMatchabilityNet

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def scale_list(nb_scale, scale_r):
    """quick_start/coarseAlignFeatMatch.py:70-74."""
    if nb_scale == 1:
        return [1]
    half = nb_scale // 2 + 1
    return np.linspace(scale_r, 1, half).tolist() + np.linspace(1, 1 / scale_r, half).tolist()[1:]


def resize_dims(w, h, min_size, mode, stride=16):
    """ResizeMaxSize (quick_start/coarseAlignFeatMatch.py:80-90) / ResizeMinSize
    (evaluation/evalHpatch/coarseAlignFeatMatch.py:90-100)."""
    f = max if mode == "max" else min
    ratio = f(w / float(min_size), h / float(min_size))
    nw, nh = int(round(w / ratio)), int(round(h / ratio))
    return nw // stride * stride, nh // stride * stride


def pil_to_tensor(pil):
    arr = np.asarray(pil, dtype=np.uint8)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous().float().div(255)


_CELL_COORDS = collections.OrderedDict()
_CELL_COORDS_MAX = 64


def cell_coords_cached(n_rows, n_cols, device):
    """cell_coords() memoised per (shape, device): the grids are constants of the map size (a dozen tiny ATen launches per
    pyramid level and step otherwise).  LRU-bounded (64 shapes: a stream of variable-size pairs must not grow it for ever).
    Captured HIP graphs (_features_graphed, prepare_and_features) bake these tensors' addresses into their kernels: the feature
    dict a capture returns carries the tensors it read (``_coord_refs``), so an evicted entry stays alive for as long as a graph
    that replays against it does."""
    key = (n_rows, n_cols, str(device))
    v = _CELL_COORDS.get(key)
    if v is None:
        v = _CELL_COORDS[key] = cell_coords(n_rows, n_cols, device)
        while len(_CELL_COORDS) > _CELL_COORDS_MAX:
            _CELL_COORDS.popitem(last=False)
    else:
        _CELL_COORDS.move_to_end(key)
    return v


def cell_coords(n_rows, n_cols, device):
    """outil.getWHTensor (utils/outil.py:21-24): (W = row coordinate, H = column coordinate).
    Evaluated on the CPU and uploaded: ``(i + 0.5) / n`` is a true IEEE division there, while ATen's device kernel for
    tensor / python-scalar multiplies by the rounded reciprocal -- 40 % of the cell centres come out one float32 ulp
    (1.2e-7) away (scripts/dbg/cell_coords_check.py).  The CPU reference is the parity target: with its coordinates the
    RANSAC inputs are the oracle's bit for bit (the 1-ulp offset used to reach H as up to 4e-7, and as 4e-6 through an
    ill-conditioned 4-point sample)."""
    r = (torch.arange(n_rows, dtype=torch.float32) + 0.5) / n_rows
    c = (torch.arange(n_cols, dtype=torch.float32) + 0.5) / n_cols
    W = r.view(-1, 1).expand(n_rows, n_cols).reshape(-1)
    Hh = c.view(1, -1).expand(n_rows, n_cols).reshape(-1)
    return ((W - 0.5) * 2).to(device), ((Hh - 0.5) * 2).to(device)


class AlignPipeline:
    def __init__(self, sds, nbScale=7, nbIter=1000, tolerance=0.05, minSize=640, scaleR=1.2, variant="A",
                 device="cuda", kernelSize=7, draw="device", seed=0, degenerate="lapack", score_chunk=None):
        """``draw``: where the RANSAC index draw of utils/outil.py:120 happens when no explicit ``samples`` / ``sample_fn`` is
        given.  "device" (default; the reference draws on ``match1.device``, i.e. on the GPU in a GPU run): Philox4x32-10 on
        the device keyed by (``seed``, call counter, pair, hypothesis) -- no host sync for nbMatch, no CPU draw + upload.
        "host": ``torch.randint`` on the CPU generator per pair in pair order = what a CPU run of the reference draws
        (the parity mode: the oracle can replay it from ``torch.manual_seed``).
        ``score_chunk``: how the mutual-NN scores are summed (ops.resolve_score_chunk, resolved HERE, once): None = RFX_SCORE_CHUNK
        or the fixed default of 256 products per chunk; an int; "host" = the K blocking of this host's sgemm (scores bit-equal to
        the reference's torch.mm on this host: the parity modes).  The resolved value and its origin are attributes
        (``score_chunk``, ``score_chunk_source``) and an explicit argument of every mutual-NN launch of THIS pipeline."""
        self.score_chunk, self.score_chunk_source = ops.resolve_score_chunk(score_chunk)
        if draw not in ("device", "host"):
            raise ValueError("draw must be 'device' or 'host'")
        if degenerate not in ("auto", "device", "lapack"):
            raise ValueError("degenerate must be 'auto', 'device' or 'lapack'")
        # rank-deficient 4-point samples (ops.ransac_h4_batched): "lapack" (default since round 6: the EXACT mode) = re-solved by the
        # host's LAPACK like the reference does for every hypothesis -- the flagged samples go to pinned memory, librfxhost.so runs
        # numpy's own dgesdd on them on std::threads, one more host wait per RANSAC call, hidden under other device work by the
        # lock-step drivers (multi_h_batched: 0.985x of the "device" mode on BASELINE config 3); "device" = the Householder sweep's own
        # null vector for those samples (no host round trip; a valid vector of the 2-D null space, not LAPACK's pick);
        # "auto" (rounds 4-5's default) = lapack whenever the draw comes from the host, device otherwise
        self.degenerate = degenerate
        if degenerate == "lapack" or (degenerate == "auto" and draw == "host"):
            from . import _lapack
            _lapack.start()           # the exact mode's host LAPACK workers: started here, not inside the first round
        self.draw, self.seed, self._draw_calls = draw, int(seed), 0
        self.dev = torch.device(device)
        self.trunk = ResNet50Trunk(sds["trunk"], self.dev)
        self.feat = FeatureExtractorNet(sds["feat"], self.dev) if "feat" in sds else None
        self.flow = NetFlowCoarseNet(sds["flow"], kernelSize, self.dev) if "flow" in sds else None
        self.match = NetMatchabilityNet(sds["match"], kernelSize, self.dev) if "match" in sds else None
        self.nbIter, self.tol, self.minSize = nbIter, tolerance, minSize
        self.variant = variant
        self.scaleList = scale_list(nbScale, scaleR)
        self.mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
        self.std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)

    def _degenerate_mode(self, host_draw):
        if self.degenerate == "auto":
            return "lapack" if host_draw else "device"
        return self.degenerate

    def reseed(self, seed):
        """Restart the device-side index-draw stream (same seed + same call sequence -> same draws)."""
        self.seed, self._draw_calls = int(seed), 0

    def _pair_draw(self, n, it):
        """Index draw of the per-pair drivers: the pipeline's draw mode for one pair with n matches."""
        if self.draw == "device":
            ids, epoch = self._draw_epoch(None)
            return self._device_draw(torch.tensor([n], dtype=torch.int32, device=self.dev), ids, epoch, 0)[0]
        return torch.randint(n, (it, 4))

    # which driver draws: part of the Philox stream id when pair_ids are given
    DRAW_TAG = dict(multi_h=0, coarse=1, kitti=2, variant_c_select=3, variant_c=4)

    def _draw_epoch(self, pair_ids, driver="multi_h", draw_epoch=0):
        """The key of a driver call's device draws besides (seed, round): ``pair_ids`` given (the caller's ABSOLUTE pair ids,
        one per pair of the batch) -> (ids on the device, epoch): a pair's hypotheses depend on (seed, its id, the driver, the
        caller's ``draw_epoch``, the round) only -- not on the batch it rides in, on which other pairs are still active, or on how
        a stream is sharded over ranks.  epoch = draw_epoch * 16 + the driver's tag: the quick_start search (coarse) and round 0 of
        a multi-homography loop work on the same match list and must not share their samples, and a caller that retries a pair
        passes another ``draw_epoch`` to draw afresh (draw_epoch 0 + multi_h = epoch 0: the keys of rounds 3-4 are unchanged).
        The id tensor is cached per id tuple and uploaded from pinned memory without blocking.
        ``None`` -> ids = positions in the batch and a per-pipeline call counter as the epoch (calls draw afresh)."""
        if pair_ids is None:
            self._draw_calls += 1
            return None, self._draw_calls
        key = tuple(int(i) for i in pair_ids)
        cache = self.__dict__.setdefault("_ids_cache", {})
        ids = cache.get(key)
        if ids is None:
            if len(cache) >= 16:
                cache.pop(next(iter(cache)))
            ids = cache[key] = torch.tensor(key, dtype=torch.int32).pin_memory().to(self.dev, non_blocking=True)
        return ids, int(draw_epoch) * 16 + self.DRAW_TAG[driver]

    def _device_draw(self, n_dev, ids=None, epoch=0, rnd=0):
        return ops.draw_samples(n_dev, self.nbIter, self.seed, (int(epoch) << 32) | int(rnd), ids)

    # ---------------------------------------------------------------- host pre-processing
    def _resize(self, I, size):
        w, h = I.size
        nw, nh = resize_dims(w, h, size, "max" if self.variant == "A" else "min")
        return I.resize((nw, nh), resample=Image.LANCZOS)

    def prepare(self, pairs):
        """pairs: list of (Is, It) PIL images, all of one size.  Returns device-resident inputs."""
        nS = len(self.scaleList)
        src = [[] for _ in range(nS)]
        tgt, IsT, ItT = [], [], []
        for Is_org, It_org in pairs:
            pyr = [self._resize(Is_org, int(self.minSize * s)) for s in self.scaleList]
            for i, im in enumerate(pyr):
                src[i].append(pil_to_tensor(im))
            IsT.append(src[nS // 2][-1])
            It = self._resize(It_org, self.minSize)
            tgt.append(pil_to_tensor(It))
        norm = lambda lst: ((torch.stack(lst) - self.mean) / self.std).to(self.dev)
        return dict(src=[norm(s) for s in src], tgt=norm(tgt), IsTensor=torch.stack(IsT).to(self.dev),
                    ItTensor=torch.stack(tgt).to(self.dev), B=len(pairs))

    def upload_raw(self, pairs):
        """Raw uint8 images of a batch -> device (N,H,W,3) tensors (what a decoder would leave in HBM)."""
        src = torch.from_numpy(np.stack([np.asarray(p[0].convert("RGB"), dtype=np.uint8) for p in pairs])).to(self.dev)
        tgt = torch.from_numpy(np.stack([np.asarray(p[1].convert("RGB"), dtype=np.uint8) for p in pairs])).to(self.dev)
        return src, tgt

    def prepare_device(self, src_u8, tgt_u8):
        """Same result as ``prepare`` -- bit for bit -- with the LANCZOS pyramid, ToTensor and Normalize computed on
        the device from raw uint8 (N,H,W,3) images: Pillow's fixed-point resampler and the float
        conversions are reproduced exactly by rfx_lanczos_pass_u8 / rfx_u8_to_f32_chw."""
        mode = "max" if self.variant == "A" else "min"
        B, h, w, _ = src_u8.shape
        srcs, IsT = [], None
        mid = len(self.scaleList) // 2
        for i, sc in enumerate(self.scaleList):
            nw, nh = resize_dims(w, h, int(self.minSize * sc), mode)
            im = ops.lanczos_resize_u8(src_u8, nw, nh)
            raw, norm = ops.u8_to_f32(im, IMAGENET_MEAN, IMAGENET_STD, want_raw=(i == mid))
            srcs.append(norm)
            if i == mid:
                IsT = raw
        ItT, tnorm = self.prepare_target_device(tgt_u8)
        return dict(src=srcs, tgt=tnorm, IsTensor=IsT, ItTensor=ItT, B=B)

    def prepare_target_device(self, tgt_u8):
        """The target half of prepare_device: raw uint8 (N,H,W,3) -> (ItTensor raw, normalised), resized like setTarget does."""
        mode = "max" if self.variant == "A" else "min"
        th, tw = tgt_u8.shape[1], tgt_u8.shape[2]
        nw, nh = resize_dims(tw, th, self.minSize, mode)
        return ops.u8_to_f32(ops.lanczos_resize_u8(tgt_u8, nw, nh), IMAGENET_MEAN, IMAGENET_STD)

    def _trunk(self, x):
        """self.trunk(x), optionally in sub-batches of RFX_TRUNK_CHUNK images (experiment knob, round 6: does a working set that fits
        the 256 MB Infinity Cache make the layer-to-layer re-reads cheaper than the smaller launches cost?).  Every sample is
        computed independently: bit-identical."""
        ch = int(os.environ.get("RFX_TRUNK_CHUNK", "0"))
        if ch <= 0 or x.shape[0] <= ch:
            return self.trunk(x)
        return torch.cat([self.trunk(x[i:i + ch]) for i in range(0, x.shape[0], ch)], dim=0)

    # ---------------------------------------------------------------- coarse stage
    def features(self, prep):
        """ResNet-50 conv4 features of every pyramid level and of the target, L2-normalised, written into
        featA (B,1024,nA) / featB (B,1024,nB) (quick_start/coarseAlignFeatMatch.py:92-125).

        Small batches (B <= 4) are launch-bound: ~900 kernel launches of a few workgroups each per pair cost more host time
        (Python + allocator + launch) than GPU time.  There the whole multi-stream trunk pass is captured ONCE per input
        shape into a HIP graph (``torch.cuda.CUDAGraph``: our ctypes launches go to torch's current stream, which is the
        capture stream) and replayed: one graph launch instead of ~900 (RFX_GRAPH=0 disables; never under ops.Profiler,
        whose per-launch events cannot be recorded into a graph)."""
        B = prep["B"]
        if B <= 4 and os.environ.get("RFX_GRAPH", "1") != "0" and ops.Profiler.active() is None:
            return self._features_graphed(prep)
        return self._features_eager(prep)

    MAX_GRAPHS = 4      # captured shapes kept (LRU): each pins a private memory pool + static input / output buffers

    def _features_graphed(self, prep):
        """HIP-graph replay of the trunk pass.  A shape is captured at its SECOND sighting (a stream of variable-size pairs
        never pays warm-up + capture for shapes it sees once) and at most MAX_GRAPHS captures are kept (LRU; an evicted
        entry releases its graph, pool and static buffers).  Warm-up, capture and replay run under the pipeline's own
        device: ``torch.cuda.graph`` captures the CURRENT device's stream, while the kernels launch on self.dev's."""
        import collections
        key = (tuple(tuple(x.shape) for x in prep["src"]), tuple(prep["tgt"].shape))
        cache = self.__dict__.setdefault("_graphs", collections.OrderedDict())
        seen = self.__dict__.setdefault("_graph_seen", collections.OrderedDict())
        ent = cache.get(key)
        if ent is None:
            if key not in seen:
                seen[key] = True
                while len(seen) > 64:
                    seen.popitem(last=False)
                return self._features_eager(prep)
            with torch.cuda.device(self.dev):
                self._features_eager(prep)                  # warm-up: lazily built state (packed weights ...) must exist
                torch.cuda.synchronize(self.dev)
                static = dict(src=[x.clone() for x in prep["src"]], tgt=prep["tgt"].clone(), B=prep["B"])
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = self._features_eager(static)
                # an empty capture (kernels launched outside the captured stream) would replay stale features for ever:
                # check once that a replay really rewrites the outputs
                out["featA"].zero_()
                g.replay()
                torch.cuda.synchronize(self.dev)
                if not bool(out["featA"].abs().sum() > 0):
                    raise RuntimeError("HIP-graph capture of the trunk pass is empty (kernels did not go to the capture stream)")
            ent = cache[key] = (g, static, out)
            while len(cache) > self.MAX_GRAPHS:
                cache.popitem(last=False)
        else:
            cache.move_to_end(key)
        g, static, out = ent
        with torch.cuda.device(self.dev):
            for d, x in zip(static["src"], prep["src"]):
                d.copy_(x)
            static["tgt"].copy_(prep["tgt"])
            g.replay()
            res = dict(out)
            res["featA"], res["featB"] = out["featA"].clone(), out["featB"].clone()   # the graph's own buffers are reused by the next replay
        return res

    def prepare_and_features(self, src_u8, tgt_u8):
        """prepare_device + features -> (prep, feats).  Small batches (B <= 4): ONE HIP graph from the raw uint8 images to the
        normalised features -- the device pyramid (24 launches of tiny kernels per pair) is as launch-bound as the trunk pass, and
        a graph that starts at the raw images needs 2 input copies instead of 8.  Same capture policy as _features_graphed (second
        sighting, LRU of MAX_GRAPHS, non-empty check); the returned ``prep`` tensors belong to the graph and are valid until its
        next replay.  Larger batches, RFX_GRAPH=0 or an active Profiler: the two eager calls."""
        import collections
        B = src_u8.shape[0]
        if not (B <= 4 and os.environ.get("RFX_GRAPH", "1") != "0" and ops.Profiler.active() is None):
            prep = self.prepare_device(src_u8, tgt_u8)
            return prep, self.features(prep)
        key = ("raw", tuple(src_u8.shape), tuple(tgt_u8.shape))
        cache = self.__dict__.setdefault("_graphs", collections.OrderedDict())
        seen = self.__dict__.setdefault("_graph_seen", collections.OrderedDict())
        ent = cache.get(key)
        if ent is None:
            if key not in seen:
                seen[key] = True
                while len(seen) > 64:
                    seen.popitem(last=False)
                prep = self.prepare_device(src_u8, tgt_u8)
                return prep, self._features_eager(prep)
            with torch.cuda.device(self.dev):
                self._features_eager(self.prepare_device(src_u8, tgt_u8))     # warm-up: lazily built state must exist
                torch.cuda.synchronize(self.dev)
                static = (src_u8.clone(), tgt_u8.clone())
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    prep = self.prepare_device(*static)
                    out = self._features_eager(prep)
                out["featA"].zero_()
                g.replay()
                torch.cuda.synchronize(self.dev)
                if not bool(out["featA"].abs().sum() > 0):
                    raise RuntimeError("HIP-graph capture of the pyramid + trunk pass is empty (kernels did not go to the capture stream)")
            ent = cache[key] = (g, static, (prep, out))
            while len(cache) > self.MAX_GRAPHS:
                cache.popitem(last=False)
        else:
            cache.move_to_end(key)
        g, static, (prep, out) = ent
        with torch.cuda.device(self.dev):
            static[0].copy_(src_u8)
            static[1].copy_(tgt_u8)
            g.replay()
            res = dict(out)
            res["featA"], res["featB"] = out["featA"].clone(), out["featB"].clone()   # the graph's own buffers are reused by the next replay
        return prep, res

    def _features_eager(self, prep):
        B = prep["B"]
        dims = [(x.shape[2] // 16, x.shape[3] // 16) for x in prep["src"]]
        nA = sum(r * c for r, c in dims)
        ldA = (nA + 3) // 4 * 4      # rows padded to 16 bytes: the mutual-NN kernel then stages with float4 loads
        featA = torch.empty((B, 1024, ldA), dtype=torch.float32, device=self.dev)
        Ws, Hs, offs = [], [], []
        off = 0
        for (r, c) in dims:
            W, Hh = cell_coords_cached(r, c, self.dev)
            Ws.append(W)
            Hs.append(Hh)
            offs.append(off)
            off += r * c
        tgt = prep["tgt"]
        # The pyramid levels are independent trunk passes.  With S > 1 streams (RFX_TRUNK_STREAMS) they are dealt to S HIP streams
        # (largest level first) so that the launch tails of one level -- the /16 maps of the small levels have few
        # workgroups per layer -- overlap with the next level's kernels; every level still writes its own columns of featA.
        # Default: one stream per level for small batches (B <= 4: a layer of one level has too few workgroups to fill 256 CUs
        # -- a single 480x640 pair drops from 15.9 to 10.6 ms), four streams otherwise; ONE stream under an ops.Profiler
        # (overlapping launches would distort the per-kernel event timing bench.py's rooflines are computed from).
        # (measured, coarse stage of 480x640 pairs: B = 1 7.73 -> 6.83 ms, B = 2 10.15 -> 9.99 ms, B = 4 16.7 -> 18.3 ms: from four
        # pairs on the per-level batches are large enough for the 8-stream form to win)
        grouped = (B <= 2 and os.environ.get("RFX_GROUPED", "1") != "0" and ops.Profiler.active() is None
                   and not os.environ.get("RFX_TRUNK_STREAMS"))
        ft_raw = None
        if grouped:
            # Small batches: the 8 images of a pair (7 pyramid levels + target) have 8 different sizes, so a layer is 8 launches
            # of 2-150 workgroups on 256 CUs and the pass is bound by one workgroup lifetime per layer AND level.  All levels
            # go through the trunk layer by layer with ONE grouped launch per kernel instance and layer (nets.forward_group:
            # blockIdx.y selects the image); bit-identical to the per-level passes.
            xs, shared = [], None
            for i, x in enumerate(prep["src"]):
                if shared is None and x.shape == tgt.shape:
                    xs.append(torch.cat((x, tgt), dim=0))           # the scale-1 level and the target: one problem of 2B images
                    shared = i
                else:
                    xs.append(x)
            if shared is None:
                xs.append(tgt)
            nch = max(1, int(os.environ.get("RFX_GROUP_CHAINS", "2")))
            if nch == 1:
                fs = self.trunk.forward_group(xs)
                for i, (r, c) in enumerate(dims):
                    f = fs[i][:B] if i == shared else fs[i]
                    ops.l2norm(f, out=featA[:, :, offs[i]:], out_batch_stride=1024 * ldA, out_chan_stride=ldA)
            else:
                # The images split into k pixel-balanced subsets, each its own grouped chain on its own stream (captured as a fork /
                # join of the graph): the launch tails of one chain overlap the other's kernels.  Single 480x640 pair: 6.08-6.35 -> 5.38-5.70 ms
                # with two chains on the same box (three: 5.7-6.3, four: 5.8), two pairs 10.22 -> 9.54 ms; each chain keeps its kernel
                # instances serial on its stream.  RFX_GROUP_CHAINS=1: one chain + the library's side streams (round 3's first form).
                order = sorted(range(len(xs)), key=lambda i: -xs[i].numel())
                chains, load = [[] for _ in range(nch)], [0] * nch
                for i in order:                      # largest first to the lighter chain (measured against "the two largest levels
                    k = load.index(min(load))        # vs the rest" 5.63 ms and alternating 5.99 ms: 5.38-5.43 ms)
                    chains[k].append(i)
                    load[k] += xs[i].numel()
                if getattr(self, "_chain_streams", None) is None or len(self._chain_streams) != nch - 1:
                    self._chain_streams = [torch.cuda.Stream(device=self.dev) for _ in range(nch - 1)]
                main_s = torch.cuda.current_stream(self.dev)
                ready_c = torch.cuda.Event()
                ready_c.record(main_s)
                fs, done_c = [None] * len(xs), []
                for k, idxs in enumerate(chains):
                    if not idxs:
                        continue
                    st = main_s if k == 0 else self._chain_streams[k - 1]
                    with torch.cuda.stream(st):
                        if k:
                            st.wait_event(ready_c)
                        # (side streams inside a chain, shared or one pool per chain, crash the process on ROCm 7.2: chains stay serial)
                        out = self.trunk.forward_group([xs[i] for i in idxs], side_streams=False)
                        for i, f in zip(idxs, out):
                            fs[i] = f
                            if i < len(dims):
                                ops.l2norm(f[:B] if i == shared else f, out=featA[:, :, offs[i]:], out_batch_stride=1024 * ldA,
                                           out_chan_stride=ldA)
                        if k:
                            ev = torch.cuda.Event()
                            ev.record(st)
                            done_c.append(ev)
                for ev in done_c:
                    main_s.wait_event(ev)
                for f in fs:
                    f.record_stream(main_s)
            ft_raw = fs[shared][B:] if shared is not None else fs[-1]
        env = os.environ.get("RFX_TRUNK_STREAMS")
        # larger batches: 4 streams (round 6, profiles/r06_stream_sweep.txt: 1 / 2 / 4 / 8 streams = 113.6 / 113.0 / 114.6 / 114.8 pairs/s
        # on config 3 with 3 lock-step groups) -- the tail of one level's layer overlaps another level's kernels
        nstream = max(1, int(env)) if env else (len(prep["src"]) if B <= 4 else min(4, len(prep["src"])))
        if ops.Profiler.active() is not None:
            nstream = 1          # per-launch event timing: overlapping streams would charge one kernel with another's time
        main = torch.cuda.current_stream(self.dev)
        if not grouped and nstream > 1 and (getattr(self, "_streams", None) is None or len(self._streams) != nstream):
            # (stream priorities for the largest levels -- the critical path of a small-batch pass -- were measured: 7.7 ->
            # 9.1-9.2 ms with one or two high-priority queues; all queues stay equal)
            self._streams = [torch.cuda.Stream(device=self.dev) for _ in range(nstream)]
        ready = torch.cuda.Event()
        if nstream > 1 and not grouped:
            ready.record(main)
        done = []
        for i, (x, (r, c)) in enumerate(zip(prep["src"], dims)):
            if grouped:
                break
            st = self._streams[i % nstream] if nstream > 1 else main
            with torch.cuda.stream(st):
                if nstream > 1:
                    st.wait_event(ready)
                if ft_raw is None and x.shape == tgt.shape:
                    # the pyramid level of scale 1 has the target's size: one trunk pass over both (twice the batch, one
                    # launch tail less per layer); every sample is computed independently, bit-identical to two passes
                    f2 = self._trunk(torch.cat((x, tgt), dim=0))
                    f, ft_raw = f2[:B], f2[B:]
                else:
                    f = self._trunk(x)
                ops.l2norm(f, out=featA[:, :, offs[i]:], out_batch_stride=1024 * ldA, out_chan_stride=ldA)
                if nstream > 1:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    done.append(ev)
        for ev in done:
            main.wait_event(ev)
        if ft_raw is not None and nstream > 1 and not grouped:
            ft_raw.record_stream(main)
        ft = ops.l2norm(ft_raw if ft_raw is not None else self.trunk(tgt))
        rt, ct = ft.shape[2], ft.shape[3]
        Wt, Ht = cell_coords_cached(rt, ct, self.dev)
        return dict(featA=featA, featB=ft.view(B, 1024, rt * ct), nA=nA, ldA=ldA, nB=rt * ct, WA=torch.cat(Ws), HA=torch.cat(Hs),
                    Wt=Wt, Ht=Ht, rt=rt, ct=ct, _coord_refs=(Ws, Hs))

    def coarse(self, prep, feats=None, samples=None, maskB=None, sample_fn=None, pair_ids=None, draw_epoch=0):
        """Per pair: mutual NN -> matches -> RANSAC.  Index draw (utils/outil.py:120): ``samples`` = list of (nbIter,4) int64
        CPU tensors, or ``sample_fn(b, nMatch, nbIter)`` -> such a tensor; otherwise the pipeline's ``draw`` mode -- "device":
        Philox on the device from the device-side match counts, ONE host sync per batch (the result records); "host":
        torch.randint on the CPU generator per pair, in pair order (what a CPU run of the reference draws), which needs the
        match counts on the host first (a second sync).  Returns a list of per-pair dicts (device tensors)."""
        feats = feats or self.features(prep)
        B = prep["B"]
        lib_out = []
        # ONE batched mutual-NN launch chain for all pairs
        idx1, idx2, cnt = self._mutual_batched(feats, B, maskB)
        host_draw = not (samples is None and sample_fn is None and self.draw == "device")
        self._last_degenerate = {}
        if not host_draw:
            ids, epoch = self._draw_epoch(pair_ids, "coarse", draw_epoch)
            smp = self._device_draw(cnt, ids, epoch, 0)
            counts, draws = None, smp
        else:
            counts = cnt.cpu().tolist()  # <- sync: the host-side index draw needs nbMatch
            draws = []
            for b in range(B):
                n = counts[b]
                if n >= 4:
                    draws.append(samples[b] if samples is not None else
                                 (sample_fn(b, n, self.nbIter) if sample_fn is not None else torch.randint(n, (self.nbIter, 4))))
                else:
                    draws.append(torch.zeros((self.nbIter, 4), dtype=torch.int64))
            if len({tuple(d.shape) for d in draws}) != 1:       # explicit draws of different lengths: one launch chain per pair
                return self._coarse_per_pair(feats, idx1, idx2, counts, draws, self._degenerate_mode(True))
            smp = torch.stack(draws).to(self.dev, non_blocking=True)
        M1, M2 = ops.gather_matches(idx1, idx2, cnt, feats["HA"], feats["WA"], feats["Ht"], feats["Wt"])
        bestH, inl, resd = ops.ransac_h4_batched(M1, M2, cnt, smp, self.tol, degenerate=self._degenerate_mode(host_draw),
                                                 info=self._last_degenerate)
        host = torch.cat((cnt[:, None], resd), dim=1).cpu().tolist()  # <- sync: result records (+ the match counts)
        for b in range(B):
            n, status, c, widx, nuniq = host[b]
            res = dict(index1=idx1[b, :n], index2=idx2[b, :n], H=None, inlier=None, samples=None, n=n)
            if n >= 4:
                res.update(match1=M1[b, :n], match2=M2[b, :n], samples=draws[b], status=status, count=c, winner=widx,
                           nUnique=nuniq)
                if status == 0:
                    res["H"], res["inlier"] = bestH[b], inl[b, :n]
            lib_out.append(res)
        return lib_out

    def _coarse_per_pair(self, feats, idx1, idx2, counts, draws, degenerate="device"):
        out = []
        for b, n in enumerate(counts):
            i1, i2 = idx1[b, :n], idx2[b, :n]
            res = dict(index1=i1, index2=i2, H=None, inlier=None, samples=None, n=n)
            if n >= 4:
                ones = torch.ones(n, dtype=torch.float32, device=self.dev)
                m1 = torch.stack((feats["HA"][i1], feats["WA"][i1], ones), dim=1)
                m2 = torch.stack((feats["Ht"][i2], feats["Wt"][i2], ones), dim=1)
                bestH, inl, r = ops.ransac_h4(m1, m2, draws[b].to(self.dev), self.tol, degenerate=degenerate)
                status, c, widx, nuniq = r.cpu().tolist()
                res.update(match1=m1, match2=m2, samples=draws[b], status=status, count=c, winner=widx, nUnique=nuniq)
                if status == 0:
                    res["H"], res["inlier"] = bestH, inl
            out.append(res)
        return out

    def _mutual_batched(self, feats, B, maskB=None):
        from . import _lib
        lib = _lib.load()
        nA, nB = feats["nA"], feats["nB"]
        ws = torch.empty(B * lib.rfx_mutual_nn_ws_bytes(nA, nB), dtype=torch.uint8, device=self.dev)
        cap = min(nA, nB)
        idx1 = torch.empty((B, cap), dtype=torch.int64, device=self.dev)
        idx2 = torch.empty((B, cap), dtype=torch.int64, device=self.dev)
        count = torch.zeros(B, dtype=torch.int32, device=self.dev)
        mask = None
        if maskB is not None:
            mask = (torch.stack(list(maskB)) if not isinstance(maskB, torch.Tensor) else maskB).float().contiguous()
        ldA = feats.get("ldA", nA)
        ops._call("rfx_mutual_nn_batched_f32", ops._one_device(feats["featA"], feats["featB"], mask), ops._p(feats["featA"]),
                  ldA, nA, 1024 * ldA, ops._p(feats["featB"]), nB, nB, 1024 * nB, 1024, ops._p(mask), ops._p(idx1),
                  ops._p(idx2), ops._p(count), ops._p(ws), B, int(self.score_chunk))
        return idx1, idx2, count

    # ---------------------------------------------------------------- fine stage
    def fine_quickstart(self, prep, Hs):
        """quick_start/align2images.py:61-97 for a batch: Hs (B,3,3) device tensor."""
        B, _, h, w = prep["ItTensor"].shape
        flowCoarse = ops.warp_grid(Hs, h, w)
        img1_coarse = ops.grid_sample(prep["IsTensor"], flowCoarse)
        f = ops.l2norm(self.feat(torch.cat((img1_coarse, prep["ItTensor"]), dim=0)))
        feat1, feat2 = f[:B], f[B:]
        corr12 = ops.corr_neigh(feat1, feat2)
        flowDown = self.flow(corr12, False)
        flow12, _, _ = ops.compose_flow(flowDown, flowCoarse, clamp=False)
        img1_fine = ops.grid_sample(prep["IsTensor"], flow12)
        return dict(flowCoarse=flowCoarse, img1_coarse=img1_coarse, feat1=feat1, feat2=feat2, corr12=corr12,
                    flowDown=flowDown, flow12=flow12, img1_fine=img1_fine)

    @staticmethod
    def _corr_both(featt, feats):
        """(2B,49,h,w) = corr(featt, feats) | corr(feats, featt) (evaluation/evalHpatch/evaluation.py:29,34).  RFX_CORR_BIDIR=0:
        the two directions as one launch over a doubled batch (each re-reads both feature maps), bit-identical."""
        if os.environ.get("RFX_CORR_BIDIR", "1") == "0":
            return ops.corr_neigh(torch.cat((featt, feats), dim=0), torch.cat((feats, featt), dim=0))
        B, _, h, w = featt.shape
        out = torch.empty((2 * B, 49, h, w), dtype=torch.float32, device=featt.device)
        ops.corr_neigh_bidir(featt, feats, out=out)
        return out

    def pred_flow_mask(self, IsTensor, featt, flowCoarse):
        """evaluation/evalHpatch/evaluation.py:23-55 (PredFlowMask) for a batch."""
        IsSample = ops.grid_sample(IsTensor, flowCoarse)
        feats = ops.l2norm(self.feat(IsSample))
        B = IsTensor.shape[0]
        # both correlation directions from ONE pass over the features (the reverse volume is the forward one at mirrored
        # taps / shifted pixels: ops.corr_neigh_bidir), both matchability passes in one batch
        c = self._corr_both(featt, feats)
        corr12 = c[:B]
        flowDown8 = self.flow(corr12, False)
        md = self.match(c, False)
        match12Down8, match21Down8 = md[:B], md[B:]
        H, W = flowCoarse.shape[1], flowCoarse.shape[2]
        match12 = ops.resize_bilinear(match12Down8, (H, W), align_corners=False)
        flow12, inb, _ = ops.compose_flow(flowDown8, flowCoarse, clamp=True, want_inb=True)
        match = match12 * inb.unsqueeze(1)
        return dict(flow12=flow12, match=match, flowDown8=flowDown8, match12Down8=match12Down8,
                    match21Down8=match21Down8)

    def pred_flow_mask_kitti(self, IsSample, ItSample, flowCoarse, out_hw=None):
        """evaluation/evalKITTI/evaluation.py:49-81: both images go through the FeatureExtractor here, and the
        matchability is cycle-checked: match = match12 * grid_sample(match21, flowUp) * in-bounds(flow12).
        ``out_hw``: resolution of the identity grid the reference passes (``grid``): the coarse grid's own by default; the
        ORIGINAL image size in the full-resolution pass of the KITTI driver (:302), where flowCoarse has the fine size."""
        B = IsSample.shape[0]
        f = ops.l2norm(self.feat(torch.cat((IsSample, ItSample), dim=0)))
        feats, featt = f[:B], f[B:]
        c = self._corr_both(featt, feats)                                                        # corr12 | corr21, one launch
        corr12 = c[:B]
        flowDown8 = self.flow(corr12, False)
        md = self.match(c, False)
        match12Down8, match21Down8 = md[:B], md[B:]
        H, W = (flowCoarse.shape[1], flowCoarse.shape[2]) if out_hw is None else out_hw
        m = ops.resize_bilinear(md, (H, W), align_corners=False)
        match12, match21 = m[:B], m[B:]
        flow12, inb, flowUp = ops.compose_flow(flowDown8, flowCoarse, clamp=True, want_inb=True, want_flow_up=True, out_hw=(H, W))
        # (match12 * cycle) * in-bounds, left to right like the reference expression (evalKITTI/evaluation.py:75-77), one kernel
        match = ops.match_score(match12[:, 0], ops.grid_sample(match21, flowUp)[:, 0], inb).unsqueeze(1)
        return dict(flow12=flow12, match=match, flowDown8=flowDown8, match12Down8=match12Down8,
                    match21Down8=match21Down8)

    # ---------------------------------------------------------------- multi-homography driver
    def multi_h(self, prep, b=0, maxCoarse=10, maskRegionTh=0.01, It_bg=None, feats=None, sample_fn=None):
        """The per-pair multi-homography loop of evaluation/evalHpatch/evaluation.py:184-243 (variant B: matches
        computed once, filtered by the explained-region mask) with every mask kept on the device: per accepted
        homography the host reads back two scalars (surviving-match count for the index draw, acceptance
        statistic) instead of round-tripping h x w masks through numpy as the reference does (:53,212,225,238).
        Returns dict(H=[(3,3)...], flowDown8=[...], matchDown8=[...], mask=final explained mask (h,w))."""
        feats = feats or self.features(prep)
        dev = self.dev
        h, w = prep["ItTensor"].shape[2], prep["ItTensor"].shape[3]
        IsT, ItT = prep["IsTensor"][b:b + 1], prep["ItTensor"][b:b + 1]
        i1, i2 = ops.mutual_nn(feats["featA"][b], feats["featB"][b], ldA=feats.get("ldA"), nA=feats["nA"], score_chunk=self.score_chunk)
        W1, H1 = feats["WA"][i1], feats["HA"][i1]
        W2, H2 = feats["Wt"][i2], feats["Ht"][i2]
        rt, ct = feats["rt"], feats["ct"]
        r2, c2 = i2 // ct, i2 % ct                       # integer cell coordinates (getWHTensor_Int)
        featt = ops.l2norm(self.feat(ItT))
        bg = torch.ones((h, w), dtype=torch.float32, device=dev) if It_bg is None else It_bg.to(dev).float()
        Mask = torch.zeros((h, w), dtype=torch.float32, device=dev)
        out = dict(H=[], flowDown8=[], matchDown8=[])
        draw = sample_fn or self._pair_draw
        nb = 0
        while nb <= maxCoarse:
            fg = ((Mask + (1 - bg)) > 0.5).float()
            keep = ops.resize_bilinear((1 - fg)[None, None], (rt, ct), align_corners=False)[0, 0] > 0.5
            valid = keep[r2, c2]
            n = int(valid.sum().item())                                       # sync: size of the index draw
            if n < 4:
                break
            ones = torch.ones(n, dtype=torch.float32, device=dev)
            m1 = torch.stack((H1[valid], W1[valid], ones), dim=1)
            m2 = torch.stack((H2[valid], W2[valid], ones), dim=1)
            bestH, inl, res = ops.ransac_h4(m1, m2, draw(n, self.nbIter).to(dev), self.tol,
                                            degenerate=self._degenerate_mode(sample_fn is not None or self.draw == "host"))
            flowCoarse = ops.warp_grid(bestH[None], h, w)
            pm = self.pred_flow_mask(IsT, featt, flowCoarse)
            stat = torch.stack(((pm["match"][0, 0] * (1 - fg)).mean(), res[0].float()))
            gain, status = stat.cpu().tolist()                                # sync: acceptance statistic + status
            if status != 0:
                break
            if gain > maskRegionTh or nb == 0:
                out["H"].append(bestH)
                out["flowDown8"].append(pm["flowDown8"])
                out["matchDown8"].append(torch.cat((pm["match12Down8"], pm["match21Down8"]), dim=1))
                nb += 1
                Mask = ((Mask + pm["match"][0, 0] * (1 - fg)) >= 1.0).float()
            else:
                break
        out["mask"] = Mask
        return out

    def _round_draws(self, active, n_dev, sample_fn, A=None, ids=None, epoch=0, rnd=0):
        """Index draws of one lock-step round -> (a,nbIter,4) int64 device tensor.  Device mode: Philox from the device-side
        counts keyed by (seed, epoch, round, pair id) -- the id of the k-th active pair is ids[active[k]] (the caller's absolute
        ids) or its position active[k] in the batch -- no sync.  Host mode / explicit sample_fn: one sync for the counts, CPU
        draws for the active pairs in ascending order, one upload."""
        if sample_fn is None and self.draw == "device":
            if ids is not None:
                key = ids if A is None else ids.index_select(0, A)
            else:
                key = A                                                   # None = the whole batch in order = positions 0..B-1
            return self._device_draw(n_dev, key, epoch, rnd)
        n_host = n_dev.cpu().tolist()                                                   # sync: sizes of the index draws
        draw = sample_fn or (lambda b, n, it: torch.randint(n, (it, 4)))
        return torch.stack([draw(b, n, self.nbIter) if n >= 4 else torch.zeros((self.nbIter, 4), dtype=torch.int64)
                            for b, n in zip(active, n_host)]).to(self.dev, non_blocking=True)

    def multi_h_batched(self, prep, maxCoarse=10, maskRegionTh=0.01, It_bg=None, feats=None, sample_fn=None, records=None,
                        want_lists=True, trace=None, pair_ids=None, draw_epoch=0, split=None):
        """multi_h() for every pair of the batch in lock-step: round k computes the k-th homography of all pairs that are
        still active.  A round is device work end to end -- rfx_filter_matches_f32 (mask -> keep map -> ordered compaction of
        the cached matches), the index draw (device mode), rfx_ransac_h4_batched, the warp, PredFlowMask over the active
        pairs, rfx_multih_accept_f32 (gain, accept rule, mask update, result-record store) -- and ONE host readback: the
        accept flags, from which the host builds the next round's active list (the exact mode, degenerate="lapack", has a second
        wait per round: the flagged 4-point samples, re-solved by the host's LAPACK).  Semantics per pair =
        evaluation/evalHpatch/evaluation.py:184-243.
        ``split``: the rounds of the batch as this many independent lock-step groups (contiguous slices of the batch), each on its
        own HIP stream, driven from this one host thread as coroutines (_multi_h_rounds yields at its host waits; _drive_rounds
        resumes the group whose event is due): while one group waits for the host -- accept flags, the LAPACK stage -- the other
        groups' kernels keep the GPU busy, and the small launches of one group's late rounds share the chip with another's.
        Per pair the arithmetic is unchanged (every kernel computes a pair independently; device draws are keyed by absolute
        pair position / id).  None = RFX_MULTIH_SPLIT or, for device draws, 3 groups from 12 pairs on, 2 from 8 (measured,
        profiles/r06_stream_sweep.txt: config 3, 64 pairs: 1 / 2 / 3 / 4 / 6 groups = 110.2 / 112.7 / 115.3 / 115.4 / 115.6 pairs/s;
        config 4, 16 pairs, 50 000 hypotheses: 2 / 3 / 4 groups = 56.1 / 57.4 / 56.2), else 1; forced to 1 with host draws /
        ``sample_fn`` (the CPU generator is consumed in pair order), with ``trace`` and under an ops.Profiler (per-launch events).
        ``sample_fn(b, n, nbIter)`` -> (nbIter,4) int64 CPU tensor: explicit draws (parity mode; costs a second sync per round).
        ``pair_ids``: absolute ids of the batch's pairs for the device draw (see _draw_epoch): with them a pair's homographies
        are the same alone, in any batch and under any sharding.
        ``records``: an ops.MultiHRecords to fill (the fixed-size rows one all_gather moves); ``want_lists=False`` skips the
        per-pair Python lists (throughput drivers that only ship the records).  ``trace``: a list that receives one dict per
        round with the round's state (mask before the round, counts, H, PredFlowMask outputs, accept flags) -- the parity
        sweeps replay every round on the oracle from it.
        Returns a list of dicts like multi_h() (H / flowDown8 / matchDown8 lists are views of per-round tensors)."""
        feats = feats or self.features(prep)
        dev = self.dev
        B = prep["B"]
        h, w = prep["ItTensor"].shape[2], prep["ItTensor"].shape[3]
        idx1, idx2, cnt = self._mutual_batched(feats, B)
        host_draw = sample_fn is not None or self.draw == "host"
        if split is None:
            split = int(os.environ.get("RFX_MULTIH_SPLIT", "0")) or (4 if B >= 32 else (3 if B >= 12 else (2 if B >= 8 else 1)))
        if host_draw or trace is not None or ops.Profiler.active() is not None and os.environ.get("RFX_MULTIH_SPLIT_PROFILED", "0") != "1":
            split = 1
        split = max(1, min(int(split), B))
        ids, epoch = self._draw_epoch(pair_ids, "multi_h", draw_epoch)
        if ids is None and split > 1:
            ids = torch.arange(B, dtype=torch.int32, device=dev)       # the key of pair b stays its batch position b
        st = dict(prep=prep, feats=feats, idx1=idx1, idx2=idx2, cnt=cnt, h=h, w=w, B=B, ids=ids, epoch=epoch,
                  bg=None if It_bg is None else It_bg.to(dev).float().contiguous(),
                  Mask=torch.zeros((B, h, w), dtype=torch.float32, device=dev), nbH=torch.zeros(B, dtype=torch.int32, device=dev),
                  outs=[dict(H=[], flowDown8=[], matchDown8=[]) for _ in range(B)], nb=[0] * B, records=records,
                  eye=torch.eye(3, device=dev), degenerate=self._degenerate_mode(host_draw),
                  featt=None if split > 1 or self._degenerate_mode(host_draw) == "lapack" else ops.l2norm(self.feat(prep["ItTensor"])))
        bounds = [(B * k // split, B * (k + 1) // split) for k in range(split)]
        gens = [self._multi_h_rounds(st, lo, hi, maxCoarse, maskRegionTh, sample_fn, want_lists, trace) for lo, hi in bounds]
        self._drive_rounds(gens)
        outs, Mask = st["outs"], st["Mask"]
        for b in range(B):
            outs[b]["mask"] = Mask[b]
            outs[b]["nbH"] = st["nb"][b]
            outs[b]["matches"] = (idx1[b], idx2[b], cnt[b:b + 1])      # the cached mutual matches (rows beyond the count: undefined)
        return outs

    def _drive_rounds(self, gens):
        """Run lock-step round generators to completion from this host thread: generator k runs under stream k (the caller's
        stream for k = 0, pipeline-owned side streams after), yields a HIP event whenever it needs the host to see device results,
        and is resumed once that event has completed -- whichever group is ready first.  One generator: the plain sequential loop."""
        main = torch.cuda.current_stream(self.dev)
        if len(gens) > 1:
            pool = self.__dict__.setdefault("_round_streams", [])
            while len(pool) < len(gens) - 1:
                pool.append(torch.cuda.Stream(device=self.dev))
            streams = [main] + pool[:len(gens) - 1]
            for s in streams[1:]:
                s.wait_stream(main)
        else:
            streams = [main]
        live = [(g, s, None) for g, s in zip(gens, streams)]
        while live:
            # the group whose event has already completed goes first (its host work -- the LAPACK stage, the next round's
            # launches -- then runs under the other groups' queued kernels); none ready: wait for the one that yielded first
            k = next((i for i, (_, _, ev) in enumerate(live) if ev is None or ev.query()), 0)
            g, s, ev = live.pop(k)
            if ev is not None:
                ev.synchronize()
            with torch.cuda.stream(s):
                try:
                    ev = next(g)
                except StopIteration:
                    continue
            live.append((g, s, ev))
        for s in streams[1:]:
            main.wait_stream(s)

    def _multi_h_rounds(self, st, lo, hi, maxCoarse, maskRegionTh, sample_fn, want_lists, trace):
        """The rounds of pairs [lo, hi) of a batch as a generator (see multi_h_batched / _drive_rounds): yields a recorded HIP event
        at every host wait.  All launches go to the stream that is current while the generator runs."""
        dev = self.dev
        prep, feats, B, h, w = st["prep"], st["feats"], st["B"], st["h"], st["w"]
        G = hi - lo
        whole = lo == 0 and hi == B
        cut = (lambda t: t) if whole else (lambda t: None if t is None else t[lo:hi])
        idx1, idx2, cnt = cut(st["idx1"]), cut(st["idx2"]), cut(st["cnt"])
        Mask, nbH, bg, ids = cut(st["Mask"]), cut(st["nbH"]), cut(st["bg"]), cut(st["ids"])
        IsT, ItT = cut(prep.get("IsTensor")), cut(prep.get("ItTensor"))
        rt, ct = feats["rt"], feats["ct"]
        R = st["records"]
        if R is not None and not whole:
            R = R.rows(lo, hi)
        featt = cut(st["featt"])
        kitti = st.get("kitti")
        if kitti is not None:
            kitti = dict(kitti, tensor_s=cut(kitti["tensor_s"]), tensor_d2=cut(kitti["tensor_d2"]), tensor_resize=cut(kitti["tensor_resize"]))
        outs, nb, eye, degen, epoch = st["outs"], st["nb"], st["eye"], st["degenerate"], st["epoch"]
        acc_host = torch.empty(G, dtype=torch.int32).pin_memory()
        active = list(range(G))
        rnd = 0
        while active:
            full = len(active) == G
            A = None if full else torch.tensor(active, dtype=torch.int32).pin_memory().to(dev, non_blocking=True)
            M1, M2, n_dev = ops.filter_matches(idx1, idx2, cnt, A, Mask, bg, rt, ct, feats["HA"], feats["WA"], feats["Ht"],
                                               feats["Wt"])
            smp = self._round_draws([lo + k for k in active], n_dev, sample_fn, A, ids, epoch, rnd)
            rnd += 1
            if degen == "lapack":
                # the exact mode: stage 1 + the flagged samples into pinned memory, then -- first round only -- the target's
                # FeatureExtractor pass (independent of the search) BEHIND the gather, so that the host's LAPACK stage runs
                # while the GPU works; the other rounds hide it under another group's kernels (split > 1)
                search = ops.ransac_h4_batched_begin(M1, M2, n_dev, smp, self.tol)
                if featt is None and kitti is None:
                    featt = ops.l2norm(self.feat(ItT))
                yield search.event
                info = {} if getattr(self, "exact_log", None) is not None else None
                bestH, inl, res = ops.ransac_h4_batched_finish(search, info=info)
                if info is not None:
                    self.exact_log.append(dict(info, round=rnd - 1, lo=lo, active=len(active)))
            else:
                if featt is None and kitti is None:
                    featt = ops.l2norm(self.feat(ItT))
                bestH, inl, res = ops.ransac_h4_batched(M1, M2, n_dev, smp, self.tol, degenerate=degen)
            Hs = torch.where((res[:, 0] == 0)[:, None, None], bestH, eye)                # failed pairs: any finite warp
            mask_before = (Mask if full else Mask.index_select(0, A)).clone() if trace is not None else None
            if kitti is None:
                flowCoarse = ops.warp_grid(Hs, h, w)
                Is = IsT if full else IsT.index_select(0, A)
                pm = self.pred_flow_mask(Is, featt if full else featt.index_select(0, A), flowCoarse)
                match, flow_d2 = pm["match"][:, 0], None
                accept, gain = ops.multih_accept(pm["match"], Mask, bg, A, res, n_dev, nbH, maskRegionTh, 0, bestH=bestH,
                                                 flowDown8=pm["flowDown8"], match12Down8=pm["match12Down8"],
                                                 match21Down8=pm["match21Down8"], records=R)
            else:
                # evaluation/evalKITTI/evaluation.py:279-336: the two-resolution fine pass, the small-component filter, accept mode 1
                selk = (lambda t: t) if full else (lambda t: t.index_select(0, A))
                flow_d2, pm, match = self.kitti_fine_round(Hs, selk(kitti["tensor_s"]), selk(kitti["tensor_d2"]), selk(kitti["tensor_resize"]),
                                                           (h, w), kitti["cc_th"], kitti["remove_small_cc"])
                accept, gain = ops.multih_accept(match, Mask, bg, A, res, n_dev, nbH, maskRegionTh, 1, bestH=bestH,
                                                 flowDown8=pm["flowDown8"], match12Down8=pm["match12Down8"],
                                                 match21Down8=pm["match21Down8"], flowD2=flow_d2, records=R)
            if trace is not None:
                tr = dict(active=[lo + k for k in active], mask_before=mask_before, n=n_dev, H=bestH, res=res, inlier=inl, pm=pm,
                          match=match, accept=accept, gain=gain, mask_after=(Mask if full else Mask.index_select(0, A)).clone(),
                          samples=smp, round=rnd - 1)
                if flow_d2 is not None:
                    tr["flowD2"] = flow_d2
                trace.append(tr)
            md2 = torch.cat((pm["match12Down8"], pm["match21Down8"]), dim=1) if want_lists else None
            acc_host[:len(active)].copy_(accept, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            yield ev                                                                    # the round's host readback: accept flags
            acc = acc_host[:len(active)].tolist()
            nxt = []
            for k, m in enumerate(active):
                if not acc[k]:
                    continue
                b = lo + m
                if want_lists:
                    outs[b]["H"].append(bestH[k])
                    if flow_d2 is not None:
                        outs[b]["flowD2"].append(flow_d2[k:k + 1])
                    outs[b]["flowDown8"].append(pm["flowDown8"][k:k + 1])
                    outs[b]["matchDown8"].append(md2[k:k + 1])
                nb[b] += 1
                if kitti is None:
                    if nb[b] <= maxCoarse:
                        nxt.append(m)
                elif st["records"] is not None and nb[b] >= st["records"].max_h:
                    # the reference's ``while True`` has no round limit; a fixed-size record has: the pair stops at the record's
                    # capacity and its record says so (status 4 = "capped by the driver": the reference might have gone on; a
                    # truncated pair is distinguishable from one that ended on the accept test)
                    st["records"].rec[b, 1] = 4.0
                else:
                    nxt.append(m)
            active = nxt

    # ---------------------------------------------------------------- YFCC / Corr driver shape (variant C)
    def pred_flow_mask_cycle(self, IsTensor, featt, flowCoarse):
        """PredFlowMask of evaluation/evalYFCC/evaluation.py:32-58 (= evalCorr/evaluation.py:29-56) for a batch: the Hpatch form
        (pred_flow_mask) with the CYCLE-CHECKED matchability -- match = match12 * grid_sample(match21, flowUp) * in-bounds(flow12)
        -- all at the target's resolution, target features handed in."""
        IsSample = ops.grid_sample(IsTensor, flowCoarse)
        feats = ops.l2norm(self.feat(IsSample))
        B = IsTensor.shape[0]
        c = self._corr_both(featt, feats)                                  # corr12 | corr21 from one pass
        flowDown8 = self.flow(c[:B], False)
        md = self.match(c, False)
        match12Down8, match21Down8 = md[:B], md[B:]
        H, W = flowCoarse.shape[1], flowCoarse.shape[2]
        m = ops.resize_bilinear(md, (H, W), align_corners=False)
        flow12, inb, flowUp = ops.compose_flow(flowDown8, flowCoarse, clamp=True, want_inb=True, want_flow_up=True)
        match = ops.match_score(m[:B, 0], ops.grid_sample(m[B:], flowUp)[:, 0], inb).unsqueeze(1)     # left to right, like the reference
        return dict(flow12=flow12, match=match, flowDown8=flowDown8, match12Down8=match12Down8, match21Down8=match21Down8)

    def multi_h_variant_c(self, src_u8, tgt_candidates_u8, maxCoarse=10, maskRegionTh=0.01, It_bg=None, sample_fn=None, records=None,
                          want_lists=True, pair_ids=None, draw_epoch=0):
        """The driver of evaluation/evalYFCC/evaluation.py:176-292 for a batch of B pairs in lock-step, everything on the device.

        ``src_u8`` (B,H,W,3) uint8: the sources; ``tgt_candidates_u8``: K tensors (B,Hk,Wk,3) -- candidate k of every pair (the
        script's four rotations of the target, :189; K = 1 = evalCorr-like use without a search).
        1. *candidate search* (:191-208): source pyramid features once; per candidate: target features, the variant-C getCoarse --
           keep map of the candidate's background (ops.keep_mask) -> mutual matching against the MASKED target features (the 0/1
           column mask of rfx_mutual_nn_batched_f32) -> RANSAC -- and nbInlier = the winner's inlier count (0 for the (None, [])
           sentinel: np.sum(InlierMask) counts distinct target cells, and mutual matches have distinct target cells); the chosen
           candidate is the FIRST maximum (np.argmax, :210).  One host sync: the (B,K) count table, from which the pairs are
           grouped by the shape of their chosen candidate.
        2. *multi-homography loop* (:238-275) per shape group: every round RE-MATCHES -- keep map of the explained-region mask ->
           masked mutual matching -> RANSAC -> warp -> pred_flow_mask_cycle -> accept rule / mask update / record store
           (rfx_multih_accept_f32 mode 0: the same rule as the Hpatch loop) -- with ONE host readback per round (accept flags).
        ``It_bg``: K tensors (B,Hk,Wk) (1 = foreground to explain, as the script's It_bg after :212) at the candidates' RESIZED
        target size, or None (all ones).  ``sample_fn(b, n, nbIter)``: explicit draws (parity mode), called per pair in ascending
        order, first for the candidates k = 0..K-1 with >= 4 matches, then per round.  Device draws are keyed by (seed, pair id,
        candidate / round) with their own driver tags.
        Returns a list of B dicts: H / flowDown8 / matchDown8 lists, mask, nbH, candidate (the chosen k), nbInlier (K counts)."""
        dev = self.dev
        B, K = src_u8.shape[0], len(tgt_candidates_u8)
        host_draw = sample_fn is not None or self.draw == "host"
        degen = self._degenerate_mode(host_draw)
        prep = self.prepare_device(src_u8, tgt_candidates_u8[0])
        feats = self.features(prep)
        featA, nA, ldA = feats["featA"], feats["nA"], feats.get("ldA", feats["nA"])
        cand = []                                                         # per candidate: ItTensor, featB, rt, ct, Wt, Ht
        for k, t8 in enumerate(tgt_candidates_u8):
            if k == 0:
                ItT, fB, rt, ct, Wt, Ht = prep["ItTensor"], feats["featB"], feats["rt"], feats["ct"], feats["Wt"], feats["Ht"]
            else:
                ItT, tn = self.prepare_target_device(t8)
                ft = ops.l2norm(self.trunk(tn))
                rt, ct = ft.shape[2], ft.shape[3]
                Wt, Ht = cell_coords_cached(rt, ct, dev)
                fB = ft.view(B, 1024, rt * ct)
            cand.append(dict(ItT=ItT, featB=fB, rt=rt, ct=ct, Wt=Wt, Ht=Ht,
                             bg=None if It_bg is None or It_bg[k] is None else It_bg[k].to(dev).float().contiguous()))

        def search(c, fA, maskB, n_pairs, draw_fn):
            """variant C's getCoarse for a batch: masked mutual matching -> matches -> draw -> RANSAC."""
            f = dict(featA=fA, featB=c["featB_sel"], nA=nA, ldA=ldA, nB=c["rt"] * c["ct"])
            idx1, idx2, cnt = self._mutual_batched(f, n_pairs, maskB)
            M1, M2 = ops.gather_matches(idx1, idx2, cnt, feats["HA"], feats["WA"], c["Ht"], c["Wt"])
            smp = draw_fn(cnt)
            bestH, inl, res = ops.ransac_h4_batched(M1, M2, cnt, smp, self.tol, degenerate=degen)
            return cnt, bestH, res

        ids, ep_sel = self._draw_epoch(pair_ids, "variant_c_select", draw_epoch)
        ep_loop = self._draw_epoch(pair_ids, "variant_c", draw_epoch)[1]
        all_b = list(range(B))
        counts = []
        for k, c in enumerate(cand):
            c["featB_sel"] = c["featB"]
            h, w = c["ItT"].shape[2], c["ItT"].shape[3]
            keep = None if c["bg"] is None else ops.keep_mask(None, c["bg"], None, c["rt"], c["ct"])
            n_k, _, res = search(c, featA, keep, B, lambda cnt, k=k: self._round_draws(all_b, cnt, sample_fn, None, ids, ep_sel, k))
            counts.append(torch.where((res[:, 0] == 0) & (n_k >= 4), res[:, 1], torch.zeros_like(res[:, 1])))
        table = torch.stack(counts, dim=1).cpu()                          # the ONE sync of the search: (B,K) inlier counts
        chosen = [int(max(range(K), key=lambda k: (int(table[b, k]), -k))) for b in range(B)]     # first maximum (np.argmax)
        outs = [dict(H=[], flowDown8=[], matchDown8=[], nbH=0, candidate=chosen[b], nbInlier=table[b].tolist()) for b in range(B)]
        if records is not None:
            records.rec[:, 3] = torch.tensor(chosen, dtype=torch.float32).to(dev)
        groups = {}
        for b in range(B):
            groups.setdefault(tuple(cand[chosen[b]]["ItT"].shape[2:]), []).append(b)
        IsT_all = prep["IsTensor"]
        eye = torch.eye(3, device=dev)
        for shape, members in groups.items():
            # the group's tensors: pair m of the group = pair members[m] of the batch with ITS chosen candidate
            G = len(members)
            gi = torch.tensor(members, dtype=torch.int64, device=dev)
            h, w = shape
            ref_c = cand[chosen[members[0]]]
            rt, ct = ref_c["rt"], ref_c["ct"]
            pick = lambda key: torch.stack([cand[chosen[b]][key][b] for b in members])
            ItT, fB = pick("ItT"), pick("featB")
            bg = None if all(cand[chosen[b]]["bg"] is None for b in members) else torch.stack(
                [cand[chosen[b]]["bg"][b] if cand[chosen[b]]["bg"] is not None else torch.ones((h, w), device=dev) for b in members])
            fA, IsT = featA.index_select(0, gi), IsT_all.index_select(0, gi)
            gids = gi.int() if ids is None else ids.index_select(0, gi)     # absolute batch positions when no ids were given
            featt = ops.l2norm(self.feat(ItT))
            Mask = torch.zeros((G, h, w), dtype=torch.float32, device=dev)
            nbH = torch.zeros(G, dtype=torch.int32, device=dev)
            R = None
            if records is not None:
                if (records.h8, records.w8) != (h // 8, w // 8):
                    raise ValueError("records were built for /8 maps of %dx%d, this group's targets give %dx%d (one MultiHRecords per "
                                     "shape group)" % (records.h8, records.w8, h // 8, w // 8))
                R = ops.MultiHRecords(G, records.h8, records.w8, dev, max_h=records.max_h)
                R.rec[:, 2:4] = records.rec.index_select(0, gi)[:, 2:4]
            nb = [0] * G
            active = list(range(G))
            rnd = 0
            gc = dict(ref_c)
            while active:
                full = len(active) == G
                A = None if full else torch.tensor(active, dtype=torch.int32).to(dev, non_blocking=True)
                sel = (lambda t: t) if full else (lambda t: t.index_select(0, A.long()))
                keep = ops.keep_mask(Mask, bg, A, rt, ct)
                gc["featB_sel"] = sel(fB)
                act_pairs = [members[m] for m in active]
                n_dev, bestH, res = search(gc, sel(fA), keep, len(active), lambda cnt: self._round_draws(
                    act_pairs, cnt, sample_fn, None, sel(gids), ep_loop, rnd))
                rnd += 1
                Hs = torch.where((res[:, 0] == 0)[:, None, None], bestH, eye)                # failed pairs: any finite warp
                pm = self.pred_flow_mask_cycle(sel(IsT), sel(featt), ops.warp_grid(Hs, h, w))
                accept, _ = ops.multih_accept(pm["match"], Mask, bg, A, res, n_dev, nbH, maskRegionTh, 0, bestH=bestH,
                                              flowDown8=pm["flowDown8"], match12Down8=pm["match12Down8"],
                                              match21Down8=pm["match21Down8"], records=R)
                md2 = torch.cat((pm["match12Down8"], pm["match21Down8"]), dim=1) if want_lists else None
                acc = accept.cpu().tolist()                                                 # the round's ONE sync
                nxt = []
                for k, m in enumerate(active):
                    if not acc[k]:
                        continue
                    if want_lists:
                        o = outs[members[m]]
                        o["H"].append(bestH[k])
                        o["flowDown8"].append(pm["flowDown8"][k:k + 1])
                        o["matchDown8"].append(md2[k:k + 1])
                    nb[m] += 1
                    if nb[m] <= maxCoarse:
                        nxt.append(m)
                active = nxt
            for m, b in enumerate(members):
                outs[b]["mask"], outs[b]["nbH"] = Mask[m], nb[m]
            if R is not None:
                records.rec.index_copy_(0, gi, R.rec)
        return outs

    # ---------------------------------------------------------------- KITTI two-resolution driver (BASELINE config 5)
    @staticmethod
    def resize_img_dims(w, h, stride, min_size):
        """outil.resizeImg (utils/outil.py:6-19): smaller side -> min_size, each side ROUNDED to a multiple of stride."""
        ratio = min(w / min_size, h / min_size)
        return int(round(w / ratio / stride) * stride), int(round(h / ratio / stride) * stride)

    def multi_h_kitti(self, src_u8, tgt_u8, fineSize=650, maskRegionTh=0.005, cc_th=0.01, It_bg=None, feats=None, prep=None,
                      sample_fn=None, remove_small_cc=None):
        """The per-pair driver of evaluation/evalKITTI/evaluation.py:222-336 for ONE pair of raw uint8 (1,H,W,3) device
        images (self must be a variant-B pipeline built with the KITTI coarse parameters: minSize = coarseSize 800,
        nbScale 3, scaleR 1.2, nbIter 50 000).  Per homography: getCoarse on the cached matches filtered by the
        explained-region mask (:273) -> homography grids at the half and the full fine resolution (:281-282) -> source
        (ORIGINAL resolution) warped to the half resolution (:284) -> PredFlowMask there (:290) -> its /8 flow composed
        with the full-resolution homography grid (:294-297) -> source warped by it (:299) -> PredFlowMask whose outputs live
        at the ORIGINAL target resolution (:302) -> small-component filter on the host, like the reference's skimage call
        (:321; on the device, ops.remove_small_cc; ``remove_small_cc(match ndarray, 0.99, cc_th)`` injects a host filter) -> accept iff
        ((match > 0.9999) outside the mask).mean() > maskRegionTh or first homography (:322) -> mask update (:333).
        The loop is the reference's ``while True``: it ends on the accept test or when fewer than 4 matches survive.
        LANCZOS resizes (outil.resizeImg) run on the device, byte-exact vs Pillow.  Host syncs per homography: match count,
        RANSAC status + accept statistic.
        Returns dict(H=[...], flowD2=[...], flowDown8=[...], matchDown8=[...], mask)."""
        dev = self.dev
        if prep is None:
            prep = self.prepare_device(src_u8, tgt_u8)
        feats = feats or self.features(prep)
        h_org, w_org = tgt_u8.shape[1], tgt_u8.shape[2]
        w_r, h_r = self.resize_img_dims(w_org, h_org, 8, fineSize)
        w_d2, h_d2 = self.resize_img_dims(w_org, h_org, 8, fineSize // 2)
        tensor_s, _ = ops.u8_to_f32(src_u8)                                              # source at its ORIGINAL size
        tensor_resize, _ = ops.u8_to_f32(ops.lanczos_resize_u8(tgt_u8, w_r, h_r))
        tensor_d2, _ = ops.u8_to_f32(ops.lanczos_resize_u8(tgt_u8, w_d2, h_d2))
        i1, i2 = ops.mutual_nn(feats["featA"][0], feats["featB"][0], ldA=feats.get("ldA"), nA=feats["nA"], score_chunk=self.score_chunk)
        W1, H1 = feats["WA"][i1], feats["HA"][i1]
        W2, H2 = feats["Wt"][i2], feats["Ht"][i2]
        rt, ct = feats["rt"], feats["ct"]
        r2, c2 = i2 // ct, i2 % ct
        bg = torch.ones((h_org, w_org), dtype=torch.float32, device=dev) if It_bg is None else It_bg.to(dev).float()
        Mask = torch.zeros((h_org, w_org), dtype=torch.float32, device=dev)
        out = dict(H=[], flowD2=[], flowDown8=[], matchDown8=[])
        draw = sample_fn or self._pair_draw
        nb = 0
        while True:
            fg = ((Mask + (1 - bg)) > 0.5).float()
            keep = ops.resize_bilinear((1 - fg)[None, None], (rt, ct), align_corners=False)[0, 0] > 0.5
            valid = keep[r2, c2]
            n = int(valid.sum().item())                                       # sync: size of the index draw
            if n < 4:
                break
            ones = torch.ones(n, dtype=torch.float32, device=dev)
            m1 = torch.stack((H1[valid], W1[valid], ones), dim=1)
            m2 = torch.stack((H2[valid], W2[valid], ones), dim=1)
            bestH, _, res = ops.ransac_h4(m1, m2, draw(n, self.nbIter).to(dev), self.tol,
                                          degenerate=self._degenerate_mode(sample_fn is not None or self.draw == "host"))
            hom_d2 = ops.warp_grid(bestH[None], h_d2, w_d2)
            hom_resize = ops.warp_grid(bestH[None], h_r, w_r)
            Is_d2 = ops.grid_sample(tensor_s, hom_d2)
            flow_d2 = self.pred_flow_mask_kitti(Is_d2, tensor_d2, hom_d2)["flowDown8"]
            flowCoarse, _, _ = ops.compose_flow(flow_d2, hom_resize, clamp=True)
            IsSample = ops.grid_sample(tensor_s, flowCoarse)
            pm = self.pred_flow_mask_kitti(IsSample, tensor_resize, flowCoarse, out_hw=(h_org, w_org))
            match = pm["match"][0, 0]
            if cc_th > 0:                                                     # :321 -- on the device unless a host filter is injected
                match = (ops.remove_small_cc(match, cc_th, 0.99) if remove_small_cc is None else
                         torch.from_numpy(remove_small_cc(match.cpu().numpy(), 0.99, cc_th)).to(dev))
            stat = torch.stack((((match > 0.9999).float() * (1 - fg)).mean(), res[0].float()))
            gain, status = stat.cpu().tolist()                                # sync: acceptance statistic + status
            if status != 0:
                break
            if gain > maskRegionTh or nb == 0:
                out["H"].append(bestH)
                out["flowD2"].append(flow_d2)
                out["flowDown8"].append(pm["flowDown8"])
                out["matchDown8"].append(torch.cat((pm["match12Down8"], pm["match21Down8"]), dim=1))
                nb += 1
                Mask = ((Mask + match * (1 - fg)) > 0.9999).float()
            else:
                break
        out["mask"] = Mask
        return out

    def kitti_fine_round(self, Hs, tensor_s, tensor_d2, tensor_resize, org_hw, cc_th=0.01, remove_small_cc=None):
        """The fine part of one round of the KITTI driver (evaluation/evalKITTI/evaluation.py:279-321) for a batch of
        homographies Hs (a,3,3): homography grids at the half and the full fine resolution -> source warped to the half
        resolution -> PredFlowMask there -> its /8 flow composed with the full-resolution homography grid -> source warped by
        it -> PredFlowMask whose outputs live at the ORIGINAL target resolution ``org_hw`` -> small-component filter
        (device: rfx_remove_small_cc_f32; or an injected host filter ``remove_small_cc(match ndarray, 0.99, cc_th)``).
        Returns (flow_d2 (a,2,hd2,wd2), PredFlowMask dict of the full-resolution pass, match (a,h_org,w_org) after the filter)."""
        h_d2, w_d2 = tensor_d2.shape[2], tensor_d2.shape[3]
        h_r, w_r = tensor_resize.shape[2], tensor_resize.shape[3]
        hom_d2 = ops.warp_grid(Hs, h_d2, w_d2)
        hom_resize = ops.warp_grid(Hs, h_r, w_r)
        Is_d2 = ops.grid_sample(tensor_s, hom_d2)
        flow_d2 = self.pred_flow_mask_kitti(Is_d2, tensor_d2, hom_d2)["flowDown8"]
        flowCoarse, _, _ = ops.compose_flow(flow_d2, hom_resize, clamp=True)
        IsSample = ops.grid_sample(tensor_s, flowCoarse)
        pm = self.pred_flow_mask_kitti(IsSample, tensor_resize, flowCoarse, out_hw=org_hw)
        match = pm["match"][:, 0]
        if cc_th > 0:                                                                   # :321
            if remove_small_cc is None:
                match = ops.remove_small_cc(match, cc_th, 0.99)
            else:                                                                       # an injected host filter: one round trip
                mh = match.cpu().numpy()
                match = torch.from_numpy(np.stack([remove_small_cc(mh[k], 0.99, cc_th) for k in range(mh.shape[0])])).to(self.dev)
        return flow_d2, pm, match

    def multi_h_kitti_batched(self, src_u8, tgt_u8, fineSize=650, maskRegionTh=0.005, cc_th=0.01, It_bg=None, sample_fn=None,
                              remove_small_cc=None, records=None, want_lists=True, trace=None, pair_ids=None, draw_epoch=0, split=None):
        """multi_h_kitti() for B pairs of ONE size in lock-step (src_u8 / tgt_u8: (B,H,W,3) uint8 on the device): round k
        computes the k-th homography of every pair that is still active -- one batched launch chain for the trunk features,
        the match filtering (rfx_filter_matches_f32), the index draw (device mode), RANSAC (rfx_ransac_h4_batched), the two
        warps and both PredFlowMask passes over the active pairs, the small-component filter on the device and the accept
        rule / mask update / record store (rfx_multih_accept_f32, mode 1), with ONE host readback per round (the accept
        flags).  Per pair the arithmetic and the accept / stop rule are multi_h_kitti()'s
        (evaluation/evalKITTI/evaluation.py:257-336); batch-1 fine passes become batch-a ones, which is where the time goes
        (the 3x3 kernels run at 0.59 of the matrix peak at batch 1, 0.8 at batch 8).
        ``sample_fn(b, n, nbIter)`` -> (nbIter,4) int64 CPU tensor: explicit draws (parity mode; a second sync per round).
        ``records``: ops.MultiHRecords built with the half-resolution /8 size (hd2, wd2); ``trace``: as in multi_h_batched.
        Returns a list of dicts like multi_h_kitti()."""
        dev = self.dev
        B, h_org, w_org = tgt_u8.shape[0], tgt_u8.shape[1], tgt_u8.shape[2]
        prep = self.prepare_device(src_u8, tgt_u8)
        feats = self.features(prep)
        w_r, h_r = self.resize_img_dims(w_org, h_org, 8, fineSize)
        w_d2, h_d2 = self.resize_img_dims(w_org, h_org, 8, fineSize // 2)
        tensor_s, _ = ops.u8_to_f32(src_u8)                                              # sources at their ORIGINAL size
        tensor_resize, _ = ops.u8_to_f32(ops.lanczos_resize_u8(tgt_u8, w_r, h_r))
        tensor_d2, _ = ops.u8_to_f32(ops.lanczos_resize_u8(tgt_u8, w_d2, h_d2))
        idx1, idx2, cnt = self._mutual_batched(feats, B)
        host_draw = sample_fn is not None or self.draw == "host"
        if split is None:
            # measured at config 5 (B = 8): two groups of 4 hide the exact mode's host stage but run the fine passes at batch 4
            # instead of 8 -- a wash (21.0 vs 21.1 pairs/s, profiles/r06_stream_sweep.txt); groups pay from 8 pairs per group on
            split = int(os.environ.get("RFX_MULTIH_SPLIT", "0")) or (2 if B >= 16 else 1)
        if host_draw or trace is not None or remove_small_cc is not None or (
                ops.Profiler.active() is not None and os.environ.get("RFX_MULTIH_SPLIT_PROFILED", "0") != "1"):
            split = 1                         # (an injected host filter is called in pair order: one group)
        split = max(1, min(int(split), B))
        ids, epoch = self._draw_epoch(pair_ids, "kitti", draw_epoch)
        if ids is None and split > 1:
            ids = torch.arange(B, dtype=torch.int32, device=dev)
        # the rounds are _multi_h_rounds' (the Hpatch driver's generator) with the KITTI fine pass / accept mode / stop rule: lock-step
        # groups on streams, ready-first scheduling, the exact mode's host stage hidden under the other groups' kernels
        st = dict(prep=dict(prep, IsTensor=None, ItTensor=None), feats=feats, idx1=idx1, idx2=idx2, cnt=cnt, h=h_org, w=w_org, B=B, ids=ids,
                  epoch=epoch, bg=None if It_bg is None else It_bg.to(dev).float().contiguous(),
                  Mask=torch.zeros((B, h_org, w_org), dtype=torch.float32, device=dev), nbH=torch.zeros(B, dtype=torch.int32, device=dev),
                  outs=[dict(H=[], flowD2=[], flowDown8=[], matchDown8=[]) for _ in range(B)], nb=[0] * B, records=records,
                  eye=torch.eye(3, device=dev), degenerate=self._degenerate_mode(host_draw), featt=None,
                  kitti=dict(tensor_s=tensor_s, tensor_d2=tensor_d2, tensor_resize=tensor_resize, cc_th=cc_th, remove_small_cc=remove_small_cc))
        bounds = [(B * k // split, B * (k + 1) // split) for k in range(split)]
        self._drive_rounds([self._multi_h_rounds(st, lo, hi, None, maskRegionTh, sample_fn, want_lists, trace) for lo, hi in bounds])
        outs = st["outs"]
        for b in range(B):
            outs[b]["mask"] = st["Mask"][b]
            outs[b]["nbH"] = st["nb"][b]
            outs[b]["matches"] = (idx1[b], idx2[b], cnt[b:b + 1])      # the cached mutual matches (rows beyond the count: undefined)
        return outs

    # ---------------------------------------------------------------- whole path
    def align_prepared(self, prep, fine=True, samples=None, feats=None, pair_ids=None, draw_epoch=0):
        res = self.coarse(prep, feats=feats, samples=samples, pair_ids=pair_ids, draw_epoch=draw_epoch)
        if fine:
            eye = torch.eye(3, device=self.dev)
            Hs = torch.stack([r["H"] if r["H"] is not None else eye for r in res])
            f = self.fine_quickstart(prep, Hs)
            for b, r in enumerate(res):
                r["flow12"] = f["flow12"][b:b + 1]
                r["img1_fine"] = f["img1_fine"][b:b + 1]
                r["flowDown"] = f["flowDown"][b:b + 1]
        return res

    def align_pairs(self, pairs, fine=True, samples=None):
        return self.align_prepared(self.prepare(pairs), fine=fine, samples=samples)
