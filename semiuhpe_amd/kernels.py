"""Tensor-level wrappers over the C ABI (include/suhpe.h).

Every function takes/returns torch CUDA tensors purely as device-memory handles: it checks dtype/contiguity,
allocates outputs from torch's caching allocator, and launches the hand-written HIP kernel on the current stream.
No arithmetic happens in torch here.  Activations are pixel-major: shape (N, H, W, C) contiguous fp32.
Conv filters keep torch's logical shape (Cout, Cin/g, R, S) in channels_last memory, i.e. physically
[Cout][R][S][Cin/g].
"""
import ctypes
import os
import json

import torch

from . import _lib
from ._lib import ConvDesc, call, ptr, workspace

ACT = {"none": 0, None: 0, "relu": 1, "relu6": 2, "silu": 3, "gelu": 4, "sigmoid": 5}


def _chk(t, name="tensor"):
    if t is None:
        return
    if not t.is_cuda:
        _lib.require_gpu()
        raise RuntimeError(f"semiuhpe_amd: {name} must live on the HIP device (got {t.device}); there is no CPU path")
    if t.dtype != torch.float32:
        raise ValueError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor, got strides {t.stride()} for shape {tuple(t.shape)}")


def filter_is_physical(w):
    """True when a (Cout, Cg, R, S) filter is stored [Cout][R][S][Cg]."""
    co, cg, r, s = w.shape
    want = (r * s * cg, 1, s * cg, cg)
    st = w.stride()
    return all(w.shape[i] == 1 or st[i] == want[i] for i in range(4))


def to_filter_layout(w):
    """Return `w` (Cout,Cg,R,S) re-strided so that memory is [Cout][R][S][Cg] (a view-compatible new tensor)."""
    co, cg, r, s = w.shape
    out = torch.empty_strided((co, cg, r, s), (r * s * cg, 1, s * cg, cg), dtype=w.dtype, device=w.device)
    out.copy_(w)
    return out


def conv_desc(x_shape, w_shape, stride, pad, groups):
    n, h, w_, cin = x_shape
    cout, cg, r, s = w_shape
    if cg * groups != cin:
        raise ValueError(f"conv: filter {tuple(w_shape)} does not match input channels {cin} with groups={groups}")
    ho = (h + 2 * pad - r) // stride + 1
    wo = (w_ + 2 * pad - s) // stride + 1
    return ConvDesc(n, h, w_, cin, ho, wo, cout, r, s, stride, pad, groups)


def _is_depthwise(d):
    return d.groups == d.cin and d.cin == d.cout and d.groups > 1


# ---------------------------------------------------------------------------------------------- convolution
# Optional live profile of the GEMM-shaped launches (bench.py): when PROFILE is a list, every dense conv / linear
# launch is bracketed by HIP events on the launch stream and recorded as
# (kernel name, algorithmic FLOPs, algorithmic bytes, start event, end event).
PROFILE = None


def conv_flops(d):
    """2 * MACs of one pass over the conv described by d (identical for fwd, dgrad and wgrad)."""
    return 2.0 * d.n * d.ho * d.wo * d.cout * (d.cin // d.groups) * d.r * d.s


def conv_bytes(d):
    """algorithmic HBM bytes of one pass: input + output activations + filter, each touched once (fp32)."""
    return 4.0 * (d.n * d.h * d.w * d.cin + d.n * d.ho * d.wo * d.cout + d.cout * (d.cin // d.groups) * d.r * d.s)


class _Prof:
    def __init__(self, d, pass_id):
        self.on = PROFILE is not None
        if self.on:
            self.name = _lib.load().suhpe_conv2d_kernel_name(ctypes.byref(d), pass_id).decode()
            self.rec = (self.name, conv_flops(d), conv_bytes(d), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
                        (pass_id, d.n, d.h, d.w, d.cin, d.cout, d.r, d.stride, d.groups))

    def __enter__(self):
        if self.on:
            self.rec[3].record()

    def __exit__(self, *a):
        if self.on:
            self.rec[4].record()
            ran = _lib.load().suhpe_last_kernel_name().decode()      # what the launcher really ran (run-time fallbacks included)
            PROFILE.append((ran or self.name,) + self.rec[1:])


# Optional live profile of the HBM-bound wrappers (tools/mem_profile.py): when MEM_PROFILE is a list every decorated call is
# bracketed by events and recorded as (wrapper name, bytes of its tensor arguments + results, shape of the first tensor, e0, e1).
MEM_PROFILE = None


def _memop(fn):
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kw):
        if MEM_PROFILE is None:
            return fn(*args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        tens = [a for a in list(args) + list(kw.values()) if torch.is_tensor(a)]
        outs = [o for o in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(o)]
        seen, nbytes = set(), 0
        for t in tens + outs:
            if t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                nbytes += t.numel() * t.element_size()
        MEM_PROFILE.append((fn.__name__, nbytes, tuple(tens[0].shape) if tens else (), e0, e1))
        return out
    return wrapped


# Tile autotuning: the first time a dense conv shape is seen in a pass, every tile of the kernel menu is timed on the real
# operands (cold: a 256 MiB memset between runs evicts L2 / MALL like the rest of a training step does) and the winner is
# pinned in the library (suhpe_conv2d_set_tile).  Winners are remembered by tile name in a small JSON database
# (semiuhpe_amd/tiles_mi355x.json ships the shapes of the BASELINE configs; SUHPE_TILE_DB_SAVE=<path> writes the merged
# database at exit), so known shapes are pinned without measuring and runs are reproducible.
# Off by default; bench.py, train.py and the agent switch it on (SUHPE_AUTOTUNE=0 keeps the built-in heuristics).
AUTOTUNE = False
_TUNE_VERBOSE = os.environ.get("SUHPE_AUTOTUNE_VERBOSE", "0") != "0"      # print every tile's time while the menu is measured
_tuned = set()
_flush = {}
_TILE_DB = None
_TILE_DB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiles_mi355x.json")


def _tile_db():
    global _TILE_DB
    if _TILE_DB is None:
        _TILE_DB = {}
        path = os.environ.get("SUHPE_TILE_DB", _TILE_DB_PATH)
        if os.path.exists(path):
            with open(path) as fh:
                _TILE_DB = dict(json.load(fh).get("tiles", {}))
        save = os.environ.get("SUHPE_TILE_DB_SAVE")
        if save:
            import atexit
            atexit.register(save_tile_db, save)
    return _TILE_DB


def save_tile_db(path):
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    with open(path, "w") as fh:
        json.dump({"device": "MI355X (gfx950)", "key": "pass,n,h,w,cin,cout,r,s,stride,pad,groups",
                   "tiles": dict(sorted(_tile_db().items()))}, fh, indent=0)


SOLO_TUNE = False          # rank 0's dry pass of SSLAgent._agree_on_tiles: the one place where a multi-rank job measures tiles


def tile_table():
    """{shape key: tile name} of every shape this process knows (shipped database + what it measured)"""
    return dict(_tile_db())


def adopt_tiles(table):
    """take another process's {shape key: tile name} table: shapes seen from now on are pinned by these names (rank 0's table on every rank)"""
    _tile_db().update(table)


def share_tiles_from_rank0():
    """every rank adopts rank 0's {shape: tile name} table (one object broadcast over the torch.distributed group: the host channel); a COLLECTIVE --
    call it where every rank passes (SSLAgent._agree_on_tiles: the first training step of each kind)"""
    import torch.distributed as dist
    box = [tile_table() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    adopt_tiles(box[0])
    return box[0]


def _tile_names(lib, pass_id):
    lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
    return [lib.suhpe_conv2d_tile_name(pass_id, i).decode() for i in range(lib.suhpe_conv2d_tile_count(pass_id))]


def _autotune(d, pass_id, run):
    key = (pass_id, d.n, d.h, d.w, d.cin, d.cout, d.r, d.s, d.stride, d.pad, d.groups)
    if key in _tuned:
        return
    _tuned.add(key)
    cg, ng = d.cin // d.groups, d.cout // d.groups
    if cg % 4 or ng % 4 or d.cin % 4 or d.cout % 4:
        return                                          # scalar-loader shapes have a single tile
    lib = _lib.load()
    names = _tile_names(lib, pass_id)
    skey = ",".join(str(v) for v in key)
    known = _tile_db().get(skey)
    if known in names and lib.suhpe_conv2d_set_tile(ctypes.byref(d), pass_id, names.index(known)) == 0:
        return
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and not SOLO_TUNE:
        # More than one rank: nothing is measured HERE.  Measuring would let the replicas pin different winners (different split-K partitions,
        # Winograd vs direct rounding: the replicas would stop holding the same bits), and agreeing on rank 0's winner needs a collective at a
        # point that the ranks do not reach together -- the autotuner fires the first time a rank sees a shape, and validation deals a ragged
        # number of batches to the ranks (a lone rank would wait in the broadcast while the others are in their gather).  The training shapes are
        # agreed on elsewhere: at the first training step of each kind -- a point every rank does reach -- rank 0 measures the menu on a dry pass
        # (SOLO_TUNE) and broadcasts its table (SSLAgent._agree_on_tiles, adopt_tiles); any other shape keeps the deterministic heuristic tile.
        return
    dev = torch.cuda.current_device()
    # this shape really has to be measured: drain the device first, so the trials do not share the GPU with the teacher pass /
    # filter-gradient GEMMs still running on the side streams (the host is inside this function: nothing new is enqueued)
    torch.cuda.synchronize()
    if dev not in _flush:
        _flush[dev] = torch.empty(1 << 26, dtype=torch.float32, device="cuda")
    best = (float("inf"), -1)
    for tile in range(len(names)):
        if lib.suhpe_conv2d_set_tile(ctypes.byref(d), pass_id, tile) != 0:
            continue
        run()
        ms = float("inf")
        for _ in range(4):                              # the minimum of a few cold runs is the stable statistic
            _flush[dev].zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            e1.synchronize()
            ms = min(ms, e0.elapsed_time(e1))
        if ms < best[0]:
            best = (ms, tile)
        if _TUNE_VERBOSE:
            print(f"autotune {skey} {names[tile]}: {ms * 1e3:.0f} us", flush=True)
    winner = best[1]
    lib.suhpe_conv2d_set_tile(ctypes.byref(d), pass_id, winner)
    _tile_db()[skey] = names[winner]


# Winograd F(2x2, 3x3) tile (csrc/conv_wino.hip): when the tile selected for a (descriptor, pass) is the Winograd one, the filter is brought
# into the Winograd domain first (one small launch: the weights move every step) and the *_wino entry points take it.  The buffer is a
# per-stream workspace: the conv that reads it is enqueued before the next layer's transform overwrites it.
def _wino_filter(d, pass_id, w):
    lib = _lib.load()
    if not lib.suhpe_conv2d_wino_pinned(ctypes.byref(d), pass_id):
        return None
    nbytes = lib.suhpe_conv2d_wino_filter_bytes(ctypes.byref(d), pass_id)
    fw = workspace("wino_fw%d" % pass_id, nbytes, w.device)
    call("suhpe_conv2d_wino_filter", ctypes.byref(d), pass_id, ptr(w), ptr(fw))
    return fw


def _autotune_fwd(d, x, w, y):
    """forward menu of a dense conv; the Winograd tile is timed the way it will run: filter transform + conv on the transformed filter"""
    def run():
        fw = _wino_filter(d, 0, w)
        if fw is not None:
            call("suhpe_conv2d_fwd_wino", ctypes.byref(d), ptr(x), ptr(w), ptr(fw), None, ptr(y), None, None, 0)
        else:
            call("suhpe_conv2d_fwd", ctypes.byref(d), ptr(x), ptr(w), None, ptr(y))
    _autotune(d, 0, run)


def conv2d_fwd(x, w, bias=None, stride=1, pad=0, groups=1):
    _chk(x, "x"); _chk(bias, "bias")
    assert filter_is_physical(w), "filter must be stored [Cout][R][S][Cin/g]"
    d = conv_desc(x.shape, w.shape, stride, pad, groups)
    y = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.float32, device=x.device)
    if _is_depthwise(d) and d.cin % 4 == 0:
        assert bias is None
        _dw_fwd(d, x, w, y)
    else:
        if AUTOTUNE:
            _autotune_fwd(d, x, w, y)
        with _Prof(d, 0):
            fw = _wino_filter(d, 0, w)
            if fw is not None:
                call("suhpe_conv2d_fwd_wino", ctypes.byref(d), ptr(x), ptr(w), ptr(fw), ptr(bias), ptr(y), None, None, 0)
            else:
                call("suhpe_conv2d_fwd", ctypes.byref(d), ptr(x), ptr(w), ptr(bias), ptr(y))
    return y


def conv2d_fwd_skip(x, w, bias, skip, rowscale=None, rows_per_sample=1):
    """pointwise conv / Linear closing a residual connection: (x w^T + bias) * rowscale[sample] + skip from the GEMM epilogue"""
    _chk(x, "x"); _chk(bias, "bias"); _chk(skip, "skip"); _chk(rowscale, "rowscale")
    assert filter_is_physical(w)
    d = conv_desc(x.shape, w.shape, 1, 0, 1)
    y = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.float32, device=x.device)
    assert skip.numel() == y.numel()
    if AUTOTUNE:
        _autotune(d, 0, lambda: call("suhpe_conv2d_fwd", ctypes.byref(d), ptr(x), ptr(w), ptr(bias), ptr(y)))
    with _Prof(d, 0):
        call("suhpe_conv2d_fwd_skip", ctypes.byref(d), ptr(x), ptr(w), ptr(bias), ptr(skip), ptr(rowscale), int(rows_per_sample), ptr(y))
    return y


@_memop
def _dw_fwd(d, x, w, y):
    call("suhpe_dwconv_fwd", ctypes.byref(d), ptr(x), ptr(w), ptr(y))


@_memop
def dwconv_bnact_fwd(x, mean, rstd, gamma, beta, act, w, stride=1, pad=1, seg=None, stats=False):
    """depthwise conv of act(BN(x)) without materialising it: x is the producer conv's raw output (N,H,W,C); seg = samples of the
    first batch segment (mean / rstd are then [2, C]); stats: also the output's BatchNorm records, result as dwconv_fwd_stats"""
    _chk(x, "x")
    c = x.shape[-1]
    d = conv_desc(x.shape, w.shape, stride, pad, c)
    assert _is_depthwise(d) and c % 4 == 0
    two = seg is not None and 0 < seg < d.n
    assert mean.numel() == rstd.numel() == (2 if two else 1) * c and mean.is_contiguous() and rstd.is_contiguous()
    y = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.float32, device=x.device)
    part = None
    if stats:
        blocks = _lib.load().suhpe_dwconv_fwd_stats_blocks(ctypes.byref(d))
        assert blocks > 0
        part = torch.empty((2 if two else 1) * blocks * c * 4, dtype=torch.float32, device=x.device)
    call("suhpe_dwconv_bnact_fwd", ctypes.byref(d), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act], int(seg) if two else 0, ptr(w),
         ptr(y), ptr(part))
    if not stats:
        return y
    return (y, part, 2 * blocks, (blocks, None)) if two else (y, part, blocks)


@_memop
def dwconv_bnact_wgrad(x, mean, rstd, gamma, beta, act, dy, w_shape, stride=1, pad=1, out=None, accumulate=False, seg=None):
    """filter gradient of the depthwise conv whose input is act(BN(x)), recomputed from x while loading"""
    _chk(x, "x"); _chk(dy, "dy")
    c = x.shape[-1]
    d = conv_desc(x.shape, w_shape, stride, pad, c)
    assert _is_depthwise(d) and c % 4 == 0
    two = seg is not None and 0 < seg < d.n
    assert mean.numel() == rstd.numel() == (2 if two else 1) * c
    if out is None:
        out = torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device)
        accumulate = False
    lib = _lib.load()
    nbytes = lib.suhpe_dwconv_wgrad_workspace_bytes(ctypes.byref(d))
    ws = workspace("wgrad", nbytes, x.device)
    call("suhpe_dwconv_bnact_wgrad", ctypes.byref(d), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act], int(seg) if two else 0, ptr(dy),
         ptr(out), int(accumulate), ptr(ws), nbytes)
    return out


@_memop
def _dw_dgrad(d, dy, w, dx):
    call("suhpe_dwconv_dgrad", ctypes.byref(d), ptr(dy), ptr(w), ptr(dx))


def dwconv_dgrad_bnsums_ok(x_shape, w_shape, stride, pad):
    """can the depthwise input-gradient kernel also reduce the producer BatchNorm's backward sums for this shape?"""
    d = conv_desc(x_shape, w_shape, stride, pad, x_shape[-1])
    return _is_depthwise(d) and bool(_lib.load().suhpe_dwconv_dgrad_bnsums_ok(ctypes.byref(d)))


@_memop
def dwconv_dgrad_bnsums(dy, w, x_shape, stride, pad, bn_x, mean, rstd, gamma, beta, act, seg_n, dgamma, dbeta):
    """depthwise input gradient dx (= gradient of act(BN(bn_x))) + that BatchNorm's backward sums from the same kernel
    (suhpe_dwconv_dgrad_bnsums): -> (dx, sums [2c] or [2, 2c]); dgamma / dbeta are accumulated into"""
    _chk(dy, "dy"); _chk(bn_x, "bn_x")
    d = conv_desc(x_shape, w.shape, stride, pad, x_shape[-1])
    c = d.cin
    two = seg_n > 0
    assert mean.numel() == rstd.numel() == (2 if two else 1) * c and tuple(bn_x.shape) == tuple(x_shape)
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    sums = torch.empty(((2, 2 * c) if two else (2 * c,)), dtype=torch.float32, device=dy.device)
    nbytes = _lib.load().suhpe_dwconv_dgrad_bnsums_workspace_bytes(ctypes.byref(d))
    ws = workspace("dw_bnsums", nbytes, dy.device)
    call("suhpe_dwconv_dgrad_bnsums", ctypes.byref(d), ptr(dy), ptr(w), ptr(dx), ptr(bn_x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act],
         int(seg_n) if two else 0, ptr(sums), ptr(dgamma), ptr(dbeta), 1, ptr(ws), nbytes)
    return dx, sums


def dwconv_bwd_fused_ok(x_shape, w_shape, stride, pad):
    """can the depthwise conv's whole backward (input gradient + producer BatchNorm sums + filter gradient) run as ONE launch for this shape?"""
    d = conv_desc(x_shape, w_shape, stride, pad, x_shape[-1])
    return _is_depthwise(d) and bool(_lib.load().suhpe_dwconv_bwd_fused_ok(ctypes.byref(d)))


@_memop
def dwconv_bwd_fused(dy, w, x_shape, stride, pad, bn_x, mean, rstd, gamma, beta, act, seg_n, dgamma, dbeta, dw, mid=None):
    """suhpe_dwconv_bwd_fused: the depthwise conv's input gradient dx (= gradient of act(BN(bn_x))), that BatchNorm's backward sums and the conv's
    filter gradient (accumulated into dw) from one staged tile of the output gradient.  mid = (y, s, dpool, chan, mean_d, rstd_d, gamma_d, beta_d,
    act_d, train_stats): `dy` is then dout of the SqueezeExcitation section and the conv's output gradient is formed on the fly (se_bn_bwd_apply's
    arithmetic, never stored).  -> (dx, sums [2c] or [2, 2c]); dgamma / dbeta / dw are accumulated into"""
    _chk(dy, "dy"); _chk(bn_x, "bn_x")
    d = conv_desc(x_shape, w.shape, stride, pad, x_shape[-1])
    c = d.cin
    two = seg_n > 0
    assert mean.numel() == rstd.numel() == (2 if two else 1) * c and tuple(bn_x.shape) == tuple(x_shape) and tuple(dy.shape) == tuple(x_shape)
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    sums = torch.empty(((2, 2 * c) if two else (2 * c,)), dtype=torch.float32, device=dy.device)
    nbytes = _lib.load().suhpe_dwconv_bwd_fused_workspace_bytes(ctypes.byref(d))
    ws = workspace("dw_bwd_fused", nbytes, dy.device)
    if mid is not None:
        y, s_, dp, chan, mean_d, rstd_d, gamma_d, beta_d, act_d, train_stats = mid
        _chk(y, "y"); _chk(s_, "s"); _chk(dp, "dpool")
        assert tuple(y.shape) == tuple(x_shape) and s_.numel() == dp.numel() == d.n * c and mean_d.numel() == rstd_d.numel() == (2 if two else 1) * c
        margs = (ptr(y), ptr(s_), ptr(dp), ptr(chan), ptr(mean_d), ptr(rstd_d), ptr(gamma_d), ptr(beta_d), ACT[act_d], int(bool(train_stats)))
    else:
        margs = (None, None, None, None, None, None, None, None, 0, 0)
    call("suhpe_dwconv_bwd_fused", ctypes.byref(d), ptr(dy), ptr(w), ptr(dx), ptr(bn_x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act],
         int(seg_n) if two else 0, ptr(sums), ptr(dgamma), ptr(dbeta), 1, ptr(dw), 1, *margs, ptr(ws), nbytes)
    return dx, sums


@_memop
def bn_act_bwd_apply_seg(x, dy, rows_a, c, mean, rstd, gamma, beta, act, sums):
    """the BatchNorm backward's apply pass with two batch segments, from sums [2, 2c] a producer reduced"""
    dx = torch.empty_like(x)
    call("suhpe_bn_act_bwd_apply_seg", ptr(x), ptr(dy), x.numel() // c, int(rows_a), c, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act], ptr(sums),
         ptr(dx))
    return dx


@_memop
def _dw_wgrad(d, x, dy, out, accumulate, ws, nbytes):
    call("suhpe_dwconv_wgrad", ctypes.byref(d), ptr(x), ptr(dy), ptr(out), int(accumulate), ptr(ws), nbytes)


def conv2d_fwd_affine(x, w, scale, shift, act, residual=None, residual_after_act=False, stride=1, pad=0, groups=1):
    """eval-mode conv -> BatchNorm -> act (-> + skip) in the GEMM epilogue; scale / shift from bn_eval_affine"""
    _chk(x, "x"); _chk(residual, "residual")
    assert filter_is_physical(w)
    d = conv_desc(x.shape, w.shape, stride, pad, groups)
    y = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.float32, device=x.device)
    assert residual is None or residual.shape == y.shape
    if AUTOTUNE:
        _autotune_fwd(d, x, w, y)
    with _Prof(d, 0):
        fw = _wino_filter(d, 0, w)
        if fw is not None:
            call("suhpe_conv2d_fwd_affine_wino", ctypes.byref(d), ptr(x), ptr(w), ptr(fw), ptr(scale), ptr(shift), ACT[act], ptr(residual),
                 int(residual_after_act), ptr(y))
        else:
            call("suhpe_conv2d_fwd_affine", ctypes.byref(d), ptr(x), ptr(w), ptr(scale), ptr(shift), ACT[act], ptr(residual),
                 int(residual_after_act), ptr(y))
    return y


def conv2d_fwd_act2(x, w, bias, act):
    """pointwise conv / Linear with the activation in the epilogue: -> (pre-activation, act(pre-activation)), one launch"""
    _chk(x, "x")
    assert filter_is_physical(w) or (w.shape[2] == 1 and w.shape[3] == 1)
    d = conv_desc(x.shape, w.shape, 1, 0, 1)
    pre = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.float32, device=x.device)
    out = torch.empty_like(pre)
    if AUTOTUNE:
        _autotune(d, 0, lambda: call("suhpe_conv2d_fwd", ctypes.byref(d), ptr(x), ptr(w), ptr(bias), ptr(pre)))
    with _Prof(d, 0):
        call("suhpe_conv2d_fwd_act2", ctypes.byref(d), ptr(x), ptr(w), ptr(bias), ACT[act], ptr(pre), ptr(out))
    return pre, out


def conv2d_dgrad_actgrad(dy, w, x_shape, pre, act):
    """input gradient of a pointwise conv / Linear whose INPUT was act(pre): dx = dgrad(dy) * act'(pre) from the GEMM epilogue"""
    _chk(dy, "dy"); _chk(pre, "pre")
    d = conv_desc(x_shape, w.shape, 1, 0, 1)
    assert tuple(pre.shape) == tuple(x_shape)
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    key = (w.data_ptr(), d.cin, d.cout, d.r, d.s, d.groups)
    wt = _WT_CACHE.get(key) if _WT_CACHE is not None else None
    if wt is None:
        if _WT_CACHE is not None:
            wt = _WT_CACHE[key] = torch.empty(w.numel() * 4, dtype=torch.uint8, device=dy.device)
            _WT_SEEN[key] = (w, (d.groups, d.cout // d.groups, d.cin // d.groups, d.r * d.s))
        else:
            wt = workspace("wt", w.numel() * 4, dy.device)
        call("suhpe_conv2d_transpose_filter", ctypes.byref(d), ptr(w), ptr(wt))
    if AUTOTUNE:
        _autotune(d, 1, lambda: call("suhpe_conv2d_dgrad", ctypes.byref(d), ptr(dy), ptr(wt), ptr(dx)))
    with _Prof(d, 1):
        call("suhpe_conv2d_dgrad_actgrad", ctypes.byref(d), ptr(dy), ptr(wt), ptr(pre), ACT[act], ptr(dx))
    return dx


def conv2d_dgrad_bnbwd(dy, w, x_shape, bn_x, mean, rstd, scale, shift, act, seg_n, dgamma, dbeta):
    """input gradient of a pointwise conv whose input was act(BN(bn_x)) (train mode): -> (g, sums) with g = dgrad(dy) * act'(BN(bn_x)), the
    gradient of the BatchNorm's output, and that BatchNorm's backward sums [2c] / [2, 2c] reduced by the same GEMM epilogue
    (suhpe_conv2d_dgrad_bnbwd); dgamma / dbeta are accumulated into.  The BatchNorm's apply pass then runs with activation 'none'."""
    _chk(dy, "dy"); _chk(bn_x, "bn_x")
    d = conv_desc(x_shape, w.shape, 1, 0, 1)
    assert tuple(bn_x.shape) == tuple(x_shape)
    c = d.cin
    two = seg_n > 0
    assert mean.numel() == rstd.numel() == scale.numel() == shift.numel() == (2 if two else 1) * c
    g = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    sums = torch.empty(((2, 2 * c) if two else (2 * c,)), dtype=torch.float32, device=dy.device)
    key = (w.data_ptr(), d.cin, d.cout, d.r, d.s, d.groups)
    wt = _WT_CACHE.get(key) if _WT_CACHE is not None else None
    if wt is None:
        if _WT_CACHE is not None:
            wt = _WT_CACHE[key] = torch.empty(w.numel() * 4, dtype=torch.uint8, device=dy.device)
            _WT_SEEN[key] = (w, (d.groups, d.cout // d.groups, d.cin // d.groups, d.r * d.s))
        else:
            wt = workspace("wt", w.numel() * 4, dy.device)
        call("suhpe_conv2d_transpose_filter", ctypes.byref(d), ptr(w), ptr(wt))
    nbytes = _lib.load().suhpe_conv2d_dgrad_bnbwd_workspace_bytes(ctypes.byref(d))
    ws = workspace("dgrad_bnbwd", nbytes, dy.device)
    with _Prof(d, 1):
        call("suhpe_conv2d_dgrad_bnbwd", ctypes.byref(d), ptr(dy), ptr(wt), ptr(bn_x), ptr(mean), ptr(rstd), ptr(scale), ptr(shift), ACT[act],
             int(seg_n) if two else 0, ptr(g), ptr(sums), ptr(dgamma), ptr(dbeta), 1, ptr(ws), nbytes)
    return g, sums


def repvgg_fuse(w3, w1, bn3, bn1, bnid, eps):
    """RepVGG deploy-mode filter and bias of one block (src/repvgg.py:104-142); bn* = (gamma, beta, running_mean, running_var)
    tensors, bnid None without an identity branch.  -> (w_eq like w3, ones (cout), b_eq (cout))"""
    assert filter_is_physical(w3) and filter_is_physical(w1)
    cout, cg = w3.shape[0], w3.shape[1]
    w_eq = torch.empty_strided(tuple(w3.shape), w3.stride(), dtype=torch.float32, device=w3.device)
    vec = torch.empty(2, cout, dtype=torch.float32, device=w3.device)
    arr = lambda bn: (ctypes.c_void_p * 4)(*[t.data_ptr() for t in bn]) if bn is not None else None
    call("suhpe_repvgg_fuse", ptr(w3), ptr(w1), arr(bn3), arr(bn1), arr(bnid), float(eps), cout, cg, ptr(w_eq), ptr(vec[0]), ptr(vec[1]))
    return w_eq, vec[0], vec[1]


def bn_eval_affine(running_mean, running_var, gamma, beta, eps):
    c = running_mean.numel()
    out = torch.empty(2, c, dtype=torch.float32, device=running_mean.device)
    call("suhpe_bn_eval_affine", ptr(running_mean), ptr(running_var), ptr(gamma), ptr(beta), c, float(eps), ptr(out[0]), ptr(out[1]))
    return out[0], out[1]


def bn_affine_from_stats(mean, rstd, gamma, beta):
    """train-mode BatchNorm statistics ([c] or [2, c]) -> (scale, shift) of the same shape: the per-channel affine map a consumer
    applies while it loads (conv2d_fwd_stats / conv2d_wgrad with xf=...)"""
    c = mean.shape[-1]
    segs = mean.numel() // c
    out = torch.empty((2,) + tuple(mean.shape), dtype=torch.float32, device=mean.device)
    call("suhpe_bn_affine_from_stats", ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), segs, c, ptr(out[0]), ptr(out[1]))
    return out[0], out[1]


def conv2d_fwd_stats(x, w, stride=1, pad=0, groups=1, seg=None, xf=None):
    """Dense conv forward that also returns BatchNorm partial sums from its epilogue: (y, partials, tiles).
    seg = samples in the first batch segment: the statistics are kept per segment and the result is
    (y, partials, tiles, (first-segment records, boundary record or None)).
    xf = (scale, shift, rowscale or None, act, hw, rows of the first segment): pointwise convs only -- the conv's input is
    act(x * scale + shift) * rowscale[sample], applied by the GEMM's loader (x itself is the producer's raw output)."""
    _chk(x, "x")
    assert filter_is_physical(w)
    d = conv_desc(x.shape, w.shape, stride, pad, groups)
    y = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.float32, device=x.device)
    if xf is not None:
        return _conv2d_fwd_stats_xf(d, x, w, y, seg, xf)
    if AUTOTUNE:
        _autotune_fwd(d, x, w, y)
    tiles = _lib.load().suhpe_conv2d_fwd_stats_tiles(ctypes.byref(d))
    if seg is not None and 0 < seg < d.n:
        tile_rows = _lib.load().suhpe_conv2d_fwd_stats_tile_rows(ctypes.byref(d))
        seg_row = seg * d.ho * d.wo
        part = torch.empty(d.cout * (tiles + 1) * 4, dtype=torch.float32, device=x.device)
        extra = part[d.cout * tiles * 4:]
        with _Prof(d, 0):
            fw = _wino_filter(d, 0, w)
            if fw is not None:
                call("suhpe_conv2d_fwd_wino", ctypes.byref(d), ptr(x), ptr(w), ptr(fw), None, ptr(y), ptr(part), ptr(extra), seg_row)
            else:
                call("suhpe_conv2d_fwd_stats_seg", ctypes.byref(d), ptr(x), ptr(w), ptr(y), ptr(part), ptr(extra), seg_row)
        straddle = seg_row % tile_rows != 0
        return y, part, tiles, (seg_row // tile_rows + int(straddle), extra if straddle else None)
    part = torch.empty(d.cout * tiles * 4, dtype=torch.float32, device=x.device)
    with _Prof(d, 0):
        fw = _wino_filter(d, 0, w)
        if fw is not None:
            call("suhpe_conv2d_fwd_wino", ctypes.byref(d), ptr(x), ptr(w), ptr(fw), None, ptr(y), ptr(part), None, 0)
        else:
            call("suhpe_conv2d_fwd_stats", ctypes.byref(d), ptr(x), ptr(w), ptr(y), ptr(part))
    return y, part, tiles


def dwconv_stats_blocks(x_shape, w_shape, stride, pad):
    """record planes' height of the depthwise forward's statistics epilogue (0: this shape has none)"""
    d = conv_desc(x_shape, w_shape, stride, pad, x_shape[-1])
    if not (_is_depthwise(d) and d.cin % 4 == 0):
        return 0
    return _lib.load().suhpe_dwconv_fwd_stats_blocks(ctypes.byref(d))


@_memop
def dwconv_fwd_stats(x, w, stride=1, pad=1, seg=None, xf=None):
    """depthwise forward + BatchNorm statistics records of its output, in conv2d_fwd_stats's result format:
    (y, part, tiles) or, with seg = samples of the first batch segment, (y, part, tiles, (first-segment records, None)).
    xf = ("bn", mean, rstd, gamma, beta, act, seg): the input is act(BN(x)) of the producer, applied while loading."""
    _chk(x, "x")
    if xf is not None:
        _, mean, rstd, gamma, beta, act, xseg = xf
        assert (xseg or None) == (seg or None)
        return dwconv_bnact_fwd(x, mean, rstd, gamma, beta, act, w, stride, pad, seg=seg, stats=True)
    c = x.shape[-1]
    d = conv_desc(x.shape, w.shape, stride, pad, c)
    blocks = _lib.load().suhpe_dwconv_fwd_stats_blocks(ctypes.byref(d))
    assert blocks > 0
    y = torch.empty((d.n, d.ho, d.wo, d.cout), dtype=torch.float32, device=x.device)
    two = seg is not None and 0 < seg < d.n
    part = torch.empty((2 if two else 1) * blocks * c * 4, dtype=torch.float32, device=x.device)
    call("suhpe_dwconv_fwd_stats", ctypes.byref(d), ptr(x), ptr(w), ptr(y), ptr(part), int(seg) if two else 0)
    if two:
        return y, part, 2 * blocks, (blocks, None)
    return y, part, blocks


def _conv2d_fwd_stats_xf(d, x, w, y, seg, xf):
    scale, shift, rowscale, act, hw, rows_a = xf
    lib = _lib.load()
    if AUTOTUNE:
        _autotune_fwd(d, x, w, y)
    tiles = lib.suhpe_conv2d_fwd_stats_tiles(ctypes.byref(d))
    two = seg is not None and 0 < seg < d.n
    seg_row = seg * d.ho * d.wo if two else 0
    assert seg_row == int(rows_a), "the input transform and the statistics epilogue share one segment boundary"
    part = torch.empty(d.cout * (tiles + (1 if two else 0)) * 4, dtype=torch.float32, device=x.device)
    extra = part[d.cout * tiles * 4:] if two else None
    with _Prof(d, 0):
        call("suhpe_conv2d_fwd_stats_xf", ctypes.byref(d), ptr(x), ptr(w), ptr(y), ptr(part), ptr(extra), int(seg_row), ptr(scale), ptr(shift),
             ptr(rowscale), ACT[act], int(hw))
    if two:
        tile_rows = lib.suhpe_conv2d_fwd_stats_tile_rows(ctypes.byref(d))
        straddle = seg_row % tile_rows != 0
        return y, part, tiles, (seg_row // tile_rows + int(straddle), extra if straddle else None)
    return y, part, tiles


# A finalizer runs ceil(c / 8) workgroups; from about 3 000 records per channel on (the 112 x 112 and 56 x 56 layers of a 160-image batch) the
# two-launch form that first folds runs of records on the whole chip is faster (62 -> 13 us at 15 680 records x 24 channels).
_FOLD_MIN_RECORDS = int(os.environ.get("SUHPE_BN_FOLD_MIN", "3072"))


@_memop
def bn_stats_from_partials(part, tiles, rows, c, eps, momentum, running_mean=None, running_var=None, nbt=None, consume=True):
    """train-mode statistics from a conv epilogue's (S, Q, k, n) records.  consume=True (what the layers pass: the records are scratch):
    from SUHPE_BN_FOLD_MIN records per channel on, `part` is OVERWRITTEN (fp64 pairs over each run's first record) by the two-launch
    folding finalizer; below the threshold, or with consume=False, `part` is left intact and the one-launch finalizer runs."""
    mean = torch.empty(c, dtype=torch.float32, device=part.device)
    rstd = torch.empty(c, dtype=torch.float32, device=part.device)
    if consume and tiles >= _FOLD_MIN_RECORDS:          # long record lists are reduced all over the chip first (the records are consumed)
        call("suhpe_bn_stats_from_partials_consume", ptr(part), int(tiles), int(tiles), None, int(rows), int(rows), c, float(eps), float(momentum),
             ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var), ptr(nbt))
        return mean, rstd
    call("suhpe_bn_stats_from_partials", ptr(part), int(tiles), int(rows), c, float(eps), float(momentum), ptr(mean), ptr(rstd),
         ptr(running_mean), ptr(running_var), ptr(nbt))
    return mean, rstd


@_memop
def bn_stats_from_partials_range(part, first, count, extra, rows, c, eps, momentum, running_mean=None, running_var=None, nbt=None):
    """statistics over the epilogue records [first, first + count) plus the optional boundary record `extra`"""
    mean = torch.empty(c, dtype=torch.float32, device=part.device)
    rstd = torch.empty(c, dtype=torch.float32, device=part.device)
    call("suhpe_bn_stats_from_partials_range", ptr(part), int(first), int(count), ptr(extra), int(rows), c, float(eps), float(momentum),
         ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var), ptr(nbt))
    return mean, rstd


# Transposed filters for the input-gradient GEMM are rebuilt per call, unless the caller declares the weights constant for a
# while: inside `with filter_cache():` (SSLAgent wraps the backward of a step, where two student passes share the weights)
# a layer's transposed filter is built once and reused.
_WT_CACHE = None


class filter_cache:
    """`with filter_cache(owner):` -- transposed filters are built once inside the region.  With an owner (the agent) the
    filters seen in one region are remembered on it, and from the next region on ALL of them are rebuilt by one batched launch
    at the region's start instead of one small launch per layer."""

    def __init__(self, owner=None):
        self.owner = owner

    def __enter__(self):
        global _WT_CACHE
        self.outer, _WT_CACHE = _WT_CACHE, (_WT_CACHE if _WT_CACHE is not None else {})
        if self.owner is not None and self.outer is None:
            reg = getattr(self.owner, "_wt_registry", None)
            if reg is None:
                reg = self.owner._wt_registry = {"entries": {}, "table": None, "built_for": 0}
            self.reg = reg
            if reg["entries"]:
                _transpose_registered(reg)
                for key, (w, wt, dims) in reg["entries"].items():
                    _WT_CACHE[key] = wt
        else:
            self.reg = None
        return self

    def __exit__(self, *exc):
        global _WT_CACHE
        if self.reg is not None and exc[0] is None:
            for key, wt in _WT_CACHE.items():              # filters transposed one by one in this region join the batch
                if key not in self.reg["entries"] and key in _WT_SEEN:
                    self.reg["entries"][key] = (_WT_SEEN[key][0], wt, _WT_SEEN[key][1])
            _WT_SEEN.clear()
        _WT_CACHE = self.outer


_WT_SEEN = {}          # key -> (weight tensor, (groups, Ng, Cg, RS)) of the transposes launched in the current region


def _transpose_registered(reg):
    entries = reg["entries"]
    if reg["table"] is None or reg["built_for"] != len(entries):
        rows, starts, total = [], [], 0
        for key, (w, wt, dims) in entries.items():
            rows.append((w.data_ptr(), wt.data_ptr()) + tuple(dims))
            starts.append(total)
            total += (w.numel() + 1023) // 1024
        starts.append(total)
        dev = next(iter(entries.values()))[1].device
        reg["table"] = (torch.tensor(rows, dtype=torch.int64).to(dev), torch.tensor(starts, dtype=torch.int64).to(dev), len(rows), total)
        reg["built_for"] = len(entries)
    table, starts, n, total = reg["table"]
    call("suhpe_transpose_filters_batched", ptr(table), ptr(starts), n, total)


def conv2d_dgrad(dy, w, x_shape, stride=1, pad=0, groups=1, skip=None):
    """input gradient; skip (dense convs): a tensor of the input's shape added in the GEMM epilogue -- the gradient that bypassed
    the block through its skip connection (dx = dgrad(dy) + skip, no separate add pass)"""
    _chk(dy, "dy"); _chk(skip, "skip")
    d = conv_desc(x_shape, w.shape, stride, pad, groups)
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    assert skip is None or tuple(skip.shape) == tuple(x_shape)
    if _is_depthwise(d) and d.cin % 4 == 0:
        assert skip is None
        _dw_dgrad(d, dy, w, dx)
    else:
        key = (w.data_ptr(), d.cin, d.cout, d.r, d.s, d.groups)
        known = (not AUTOTUNE) or ((1, d.n, d.h, d.w, d.cin, d.cout, d.r, d.s, d.stride, d.pad, d.groups) in _tuned)
        if known and _lib.load().suhpe_conv2d_wino_pinned(ctypes.byref(d), 1):      # Winograd tile: the filter transform reads w as stored, no transposed copy
            with _Prof(d, 1):
                fw = _wino_filter(d, 1, w)
                call("suhpe_conv2d_dgrad_wino", ctypes.byref(d), ptr(dy), None, ptr(fw), ptr(skip), ptr(dx))
            return dx
        wt = _WT_CACHE.get(key) if _WT_CACHE is not None else None
        if wt is None:
            if _WT_CACHE is not None:
                wt = _WT_CACHE[key] = torch.empty(w.numel() * 4, dtype=torch.uint8, device=dy.device)
                _WT_SEEN[key] = (w, (d.groups, d.cout // d.groups, d.cin // d.groups, d.r * d.s))
            else:
                wt = workspace("wt", w.numel() * 4, dy.device)
            call("suhpe_conv2d_transpose_filter", ctypes.byref(d), ptr(w), ptr(wt))
        if AUTOTUNE:
            def run():
                fw_ = _wino_filter(d, 1, w)
                if fw_ is not None:
                    call("suhpe_conv2d_dgrad_wino", ctypes.byref(d), ptr(dy), ptr(wt), ptr(fw_), None, ptr(dx))
                else:
                    call("suhpe_conv2d_dgrad", ctypes.byref(d), ptr(dy), ptr(wt), ptr(dx))
            _autotune(d, 1, run)
        fw = _wino_filter(d, 1, w)
        with _Prof(d, 1):
            if fw is not None:
                call("suhpe_conv2d_dgrad_wino", ctypes.byref(d), ptr(dy), ptr(wt), ptr(fw), ptr(skip), ptr(dx))
            else:
                call("suhpe_conv2d_dgrad_add", ctypes.byref(d), ptr(dy), ptr(wt), ptr(skip), ptr(dx))
    return dx


# Deferred filter-gradient reduction: inside `with deferred_wgrads():` (one backward pass, layers.NetFunction) every dense
# wgrad keeps its split-K slabs in a buffer of its own and ONE launch at the end adds them all into the gradients
# (162 small reductions per step become 2).  Same summation order as the immediate form: bit-identical results.
_DEFER = None


# Inside a deferred_wgrads() region the filter-gradient GEMMs are also moved to a second stream: a layer's wgrad depends only
# on its saved input and its output gradient, nothing on the backward chain waits for it, and it is MFMA-bound while the
# BatchNorm / activation backward kernels it then runs beside are HBM-bound.  The region joins the stream at its end.
WGRAD_OVERLAP = os.environ.get("SUHPE_WGRAD_OVERLAP", "1") != "0"
_side_streams = {}
_side_used = False


def _side_stream(device):
    key = torch.device(device)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=key)
    return _side_streams[key]


class _on_side_stream:
    """run the body on the side stream after everything enqueued so far on the current stream; keeps the given tensors alive
    for the side stream (allocator bookkeeping)"""

    def __init__(self, device, *tensors):
        self.side = _side_stream(device)
        self.tensors = tensors

    def __enter__(self):
        global _side_used
        _side_used = True
        self.side.wait_stream(torch.cuda.current_stream())
        for t in self.tensors:
            t.record_stream(self.side)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


class _same_stream:
    def __enter__(self):
        pass

    def __exit__(self, *exc):
        return False


class deferred_wgrads:
    def __enter__(self):
        global _DEFER
        self.outer, _DEFER = _DEFER, (_DEFER if _DEFER is not None else [])
        return self

    def __exit__(self, *exc):
        global _DEFER, _side_used
        jobs, _DEFER = _DEFER, self.outer
        if self.outer is not None:
            return
        used, _side_used = _side_used, False
        if exc[0] is None and jobs:
            if used:
                with torch.cuda.stream(_side_stream(jobs[0][0].device)):
                    flush_wgrads(jobs)
            else:
                flush_wgrads(jobs)
        if used:                                        # filter gradients are complete before anything that follows
            for side in _side_streams.values():
                torch.cuda.current_stream(side.device).wait_stream(side)


def flush_deferred_now():
    """Inside a deferred_wgrads() region: reduce the split-K slabs collected so far NOW (on the stream the filter-gradient GEMMs
    run on), so that the gradients of the layers already walked are final -- the bucketed gradient exchange sends a bucket as soon
    as the backward has left it.  Same per-layer summation order as the flush at the region's end: bit-identical results."""
    if _DEFER:
        if _side_used:
            with torch.cuda.stream(_side_stream(_DEFER[0][0].device)):
                flush_wgrads(_DEFER)
        else:
            flush_wgrads(_DEFER)


def backward_streams(device):
    """the streams gradient kernels of the running backward pass may be on (current stream + the filter-gradient side stream)"""
    out = [torch.cuda.current_stream(device)]
    side = _side_streams.get(torch.device(device))
    if side is not None:
        out.append(side)
    return out


_flush_ring = []          # pinned staging buffers for the job tables: (table, starts, event), used round-robin
_flush_turn = 0


def flush_wgrads(jobs):
    global _flush_turn
    dev = jobs[0][0].device
    n = len(jobs)
    if len(_flush_ring) < 4:
        # job table (1024 x 4) and chunk starts (1025) in ONE pinned staging buffer: one upload per flush
        _flush_ring.append([torch.empty(1024 * 4 + 1025, dtype=torch.int64).pin_memory(), None, None])
    slot = _flush_ring[_flush_turn % len(_flush_ring)]
    _flush_turn += 1
    if slot[2] is not None:
        slot[2].synchronize()                       # the upload that last used this staging buffer has run (normally long ago)
    assert n <= 1024
    stage = slot[0]
    total = 0
    rows, st = [], []
    for slab, out, count, splits in jobs:
        rows.append((slab.data_ptr(), out.data_ptr(), count, splits))
        st.append(total)
        total += (count + 1023) // 1024
    st.append(total)
    stage[:4 * n] = torch.tensor(rows, dtype=torch.int64).reshape(-1)
    stage[4 * n:5 * n + 1] = torch.tensor(st, dtype=torch.int64)
    both = stage[:5 * n + 1].to(dev, non_blocking=True)
    table_d, starts_d = both[:4 * n], both[4 * n:]
    slot[2] = torch.cuda.Event()
    slot[2].record()
    call("suhpe_wgrad_reduce_batched", ptr(table_d), ptr(starts_d), n, total)
    jobs.clear()                                     # the slabs go back to the caching allocator (same stream: safe to reuse)


def conv2d_wgrad(x, dy, w_shape, stride=1, pad=0, groups=1, out=None, accumulate=False, xf=None):
    """Filter gradient in the filter's physical layout.  out: tensor with the filter's shape/strides.
    xf: the forward's input transform (conv2d_fwd_stats): the conv's input is recomputed from x by the loader."""
    _chk(x, "x"); _chk(dy, "dy")
    d = conv_desc(x.shape, w_shape, stride, pad, groups)
    if out is None:
        co, cg, r, s = w_shape
        out = torch.empty_strided(tuple(w_shape), (r * s * cg, 1, s * cg, cg), dtype=torch.float32, device=x.device)
        accumulate = False
    assert filter_is_physical(out)
    lib = _lib.load()
    side = _DEFER is not None and accumulate and WGRAD_OVERLAP
    held = [t for t in (xf or ()) if torch.is_tensor(t)]        # the transform's vectors are read on the side stream too
    with (_on_side_stream(x.device, x, dy, *held) if side else _same_stream()):
        _conv2d_wgrad(lib, d, x, dy, out, accumulate, xf)
    return out


def _wgrad_partial(d, x, dy, ws, nbytes, xf):
    if xf is None:
        call("suhpe_conv2d_wgrad_partial", ctypes.byref(d), ptr(x), ptr(dy), ptr(ws), nbytes)
    else:
        scale, shift, rowscale, act, hw, rows_a = xf
        call("suhpe_conv2d_wgrad_partial_xf", ctypes.byref(d), ptr(x), ptr(dy), ptr(ws), nbytes, ptr(scale), ptr(shift), ptr(rowscale), ACT[act],
             int(hw), int(rows_a))


def _conv2d_wgrad(lib, d, x, dy, out, accumulate, xf=None):
    if _is_depthwise(d) and d.cin % 4 == 0:
        nbytes = lib.suhpe_dwconv_wgrad_workspace_bytes(ctypes.byref(d))
        ws = workspace("wgrad", nbytes, x.device)
        if xf is None:
            _dw_wgrad(d, x, dy, out, accumulate, ws, nbytes)
        else:                                           # ("bn", mean, rstd, gamma, beta, act, samples of the first segment)
            _, mean, rstd, gamma, beta, act, seg = xf
            dwconv_bnact_wgrad(x, mean, rstd, gamma, beta, act, dy, tuple(out.shape), d.stride, d.pad, out=out, accumulate=accumulate, seg=seg)
    else:
        if AUTOTUNE:
            def trial():                                # whole wgrad (split-K GEMM + slab reduction) into a scratch filter
                nb = lib.suhpe_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
                call("suhpe_conv2d_wgrad", ctypes.byref(d), ptr(x), ptr(dy), ptr(workspace("wgrad_tune", out.numel() * 4, x.device)), 0,
                     ptr(workspace("wgrad", nb, x.device)), nb)
            _autotune(d, 2, trial)
        nbytes = lib.suhpe_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
        defer = _DEFER is not None and accumulate
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if defer else workspace("wgrad", nbytes, x.device)
        with _Prof(d, 2):           # the profile brackets the split-K GEMM launch alone
            _wgrad_partial(d, x, dy, ws, nbytes, xf)
        if defer:
            _DEFER.append((ws, out, out.numel(), lib.suhpe_conv2d_wgrad_splits(ctypes.byref(d))))
        else:
            call("suhpe_conv2d_wgrad_reduce", ctypes.byref(d), ptr(ws), ptr(out), int(accumulate))
    return out


@_memop
def bias_grad(dy, c, out=None, accumulate=False):
    _chk(dy, "dy")
    rows = dy.numel() // c
    if out is None:
        out = torch.empty(c, dtype=torch.float32, device=dy.device)
        accumulate = False
    nbytes = _lib.load().suhpe_bn_workspace_bytes(c)
    ws = workspace("bn", nbytes, dy.device)
    call("suhpe_bias_grad", ptr(dy), rows, c, ptr(out), int(accumulate), ptr(ws))
    return out


# ---------------------------------------------------------------------------------------------- normalisation
@_memop
def bn_stats(x, c, eps, momentum, running_mean=None, running_var=None, nbt=None):
    _chk(x, "x")
    rows = x.numel() // c
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    rstd = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = workspace("bn", _lib.load().suhpe_bn_workspace_bytes(c), x.device)
    call("suhpe_bn_stats", ptr(x), rows, c, float(eps), float(momentum), ptr(mean), ptr(rstd), ptr(running_mean),
         ptr(running_var), ptr(nbt), ptr(ws))
    return mean, rstd


# ---- two batch segments per launch (layers.batch_segments): rows [0, rows_a) / [rows_a, rows); statistics are [2, c] tensors
@_memop
def bn_stats_seg(x, rows_a, c, eps, momentum, running_mean=None, running_var=None, nbt=None):
    _chk(x, "x")
    stats = torch.empty((2, 2, c), dtype=torch.float32, device=x.device)          # [mean | rstd][segment][c]
    ws = workspace("bn", _lib.load().suhpe_bn_workspace_bytes(c), x.device)
    call("suhpe_bn_stats_seg", ptr(x), x.numel() // c, int(rows_a), c, float(eps), float(momentum), ptr(stats[0]), ptr(stats[1]),
         ptr(running_mean), ptr(running_var), ptr(nbt), ptr(ws))
    return stats[0], stats[1]


@_memop
def bn_stats_from_partials_seg(part, tiles, first_records, boundary, rows, rows_a, c, eps, momentum, running_mean=None, running_var=None,
                               nbt=None, consume=True):
    """two-segment form of bn_stats_from_partials; same `consume` contract (`part` is overwritten from SUHPE_BN_FOLD_MIN records on)"""
    stats = torch.empty((2, 2, c), dtype=torch.float32, device=part.device)
    if consume and tiles >= _FOLD_MIN_RECORDS:
        call("suhpe_bn_stats_from_partials_consume", ptr(part), int(tiles), int(first_records), ptr(boundary), int(rows), int(rows_a), c, float(eps),
             float(momentum), ptr(stats[0]), ptr(stats[1]), ptr(running_mean), ptr(running_var), ptr(nbt))
        return stats[0], stats[1]
    call("suhpe_bn_stats_from_partials_seg", ptr(part), int(tiles), int(first_records), ptr(boundary), int(rows), int(rows_a), c, float(eps),
         float(momentum), ptr(stats[0]), ptr(stats[1]), ptr(running_mean), ptr(running_var), ptr(nbt))
    return stats[0], stats[1]


@_memop
def bn_act_fwd_seg(x, rows_a, c, mean, rstd, gamma, beta, act, residual=None, post=False, rowscale=None, hw=1):
    _chk(x, "x"); _chk(residual, "residual"); _chk(rowscale, "rowscale")
    y = torch.empty_like(x)
    call("suhpe_bn_act_fwd_seg", ptr(x), x.numel() // c, int(rows_a), c, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act], ptr(residual),
         int(bool(post)), ptr(rowscale), int(hw), ptr(y))
    return y


def bn_act_bwd_sums(x, dy, c, mean, rstd, gamma, beta, act, dgamma=None, dbeta=None, accumulate=False, y_out=None, rowscale=None, hw=1):
    """first half of the BatchNorm backward: per-channel (sum g, sum g*xhat) of the local rows -> sums [2c] (interleaved)"""
    _chk(x, "x"); _chk(dy, "dy")
    sums = torch.empty(2 * c, dtype=torch.float32, device=x.device)
    ws = workspace("bn", _lib.load().suhpe_bn_workspace_bytes(c), x.device)
    call("suhpe_bn_act_bwd_sums", ptr(x), ptr(dy), ptr(y_out), x.numel() // c, c, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act],
         ptr(rowscale), int(hw), ptr(sums), ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws))
    return sums


def bn_act_bwd_apply(x, dy, c, mean, rstd, gamma, beta, act, sums, total_rows, y_out=None, rowscale=None, hw=1):
    """second half: dx from the (possibly all-reduced) sums and the row count they cover"""
    dx = torch.empty_like(x)
    call("suhpe_bn_act_bwd_apply", ptr(x), ptr(dy), ptr(y_out), x.numel() // c, int(total_rows), c, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
         ACT[act], ptr(sums), ptr(rowscale), int(hw), ptr(dx))
    return dx


@_memop
def bn_act_pool_fwd(x, rows_a, c, mean, rstd, gamma, beta, act, hw):
    """-> (act(BN(x)), per-sample channel means of it); rows_a = 0: one segment"""
    _chk(x, "x")
    y = torch.empty_like(x)
    pool = torch.empty((x.numel() // (c * hw), c), dtype=torch.float32, device=x.device)
    call("suhpe_bn_act_pool_fwd", ptr(x), x.numel() // c, int(rows_a), c, int(hw), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act], ptr(y),
         ptr(pool))
    return y, pool


@_memop
def bn_act_pool(x, rows_a, c, mean, rstd, gamma, beta, act, hw):
    """squeeze alone: per-sample channel means of act(BN(x)); nothing but the (n, c) means is written"""
    _chk(x, "x")
    pool = torch.empty((x.numel() // (c * hw), c), dtype=torch.float32, device=x.device)
    call("suhpe_bn_act_pool_fwd", ptr(x), x.numel() // c, int(rows_a), c, int(hw), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act], None,
         ptr(pool))
    return pool


@_memop
def bn_act_scale_fwd(x, rows_a, c, mean, rstd, gamma, beta, act, s, hw):
    """act(BN(x)) * s[n, c] (SqueezeExcitation's scale applied to a BatchNorm output that is never stored)"""
    _chk(x, "x"); _chk(s, "s")
    y = torch.empty_like(x)
    call("suhpe_bn_act_scale_fwd", ptr(x), x.numel() // c, int(rows_a), c, int(hw), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act], ptr(s), ptr(y))
    return y


@_memop
def se_bn_bwd_reduce(x, dout, rows_a, c, mean, rstd, gamma, beta, act, hw):
    """-> sums (5, n, c): per-sample sums the excitation's and the BatchNorm's backward need (sums[0] = ds)"""
    _chk(x, "x"); _chk(dout, "dout")
    n = x.numel() // (c * hw)
    sums = torch.empty((5, n, c), dtype=torch.float32, device=x.device)
    call("suhpe_se_bn_bwd_reduce", ptr(x), ptr(dout), x.numel() // c, int(rows_a), c, int(hw), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act], ptr(sums))
    return sums


def se_bn_bwd_combine(sums, s, dpool, n_a, hw, dgamma, dbeta, accumulate=True):
    n, c = s.shape
    chan = torch.empty((2, 2 * c), dtype=torch.float32, device=s.device)
    call("suhpe_se_bn_bwd_combine", ptr(sums), ptr(s), ptr(dpool), n, int(n_a), c, int(hw), ptr(chan), ptr(dgamma), ptr(dbeta), int(accumulate))
    return chan


@_memop
def se_bn_bwd_apply(x, dout, rows_a, c, mean, rstd, gamma, beta, act, s, dpool, chan, train_stats, hw):
    dy = torch.empty_like(x)
    call("suhpe_se_bn_bwd_apply", ptr(x), ptr(dout), x.numel() // c, int(rows_a), c, int(hw), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act],
         ptr(s), ptr(dpool), ptr(chan), int(bool(train_stats)), ptr(dy))
    return dy


@_memop
def bn_act_bwd_seg(x, dy, rows_a, c, mean, rstd, gamma, beta, act, dgamma=None, dbeta=None, accumulate=False, y_out=None, rowscale=None, hw=1):
    _chk(x, "x"); _chk(dy, "dy")
    dx = torch.empty_like(x)
    ws = workspace("bn", _lib.load().suhpe_bn_workspace_bytes(c), x.device)
    call("suhpe_bn_act_bwd_seg", ptr(x), ptr(dy), ptr(y_out), x.numel() // c, int(rows_a), c, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
         ACT[act], ptr(rowscale), int(hw), ptr(dx), ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws))
    return dx


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


@_memop
def repvgg_bn_fwd(xs, means, rstds, gammas, betas, rows_a, c, emit_stats=False):
    """y = relu(sum_j BN_j(xs[j])) for the 2 or 3 branches of a RepVGG block in one pass; statistics [c] or [2, c] (rows_a > 0).
    emit_stats: -> (y, partials) with the BatchNorm statistics records of y in the form _BatchNorm.train_stats(partials=...) takes: the next
    block's identity BatchNorm normalises y and then needs no statistics pass of its own"""
    for t in xs:
        _chk(t, "x")
    y = torch.empty_like(xs[0])
    rows = xs[0].numel() // c
    if emit_stats:
        first = ctypes.c_int32(0)
        tiles = _lib.load().suhpe_repvgg_bn_fwd_stats_blocks(rows, int(rows_a), c, ctypes.byref(first))
        part = torch.empty((tiles, c, 4), dtype=torch.float32, device=y.device)
        call("suhpe_repvgg_bn_fwd_stats", _ptr_array(xs), _ptr_array(means), _ptr_array(rstds), _ptr_array(gammas), _ptr_array(betas), len(xs),
             rows, int(rows_a), c, ptr(y), ptr(part))
        return y, ((part, tiles, (first.value, None)) if rows_a else (part, tiles))
    call("suhpe_repvgg_bn_fwd", _ptr_array(xs), _ptr_array(means), _ptr_array(rstds), _ptr_array(gammas), _ptr_array(betas), len(xs),
         rows, int(rows_a), c, ptr(y))
    return y


@_memop
def repvgg_bn_bwd(xs, means, rstds, gammas, dout, y, rows_a, c, dgammas, dbetas, betas=None):
    """-> [dx_j]: BatchNorm backward of every branch on the shared g = dout * (y > 0); dgammas / dbetas are accumulated.
    betas: the ReLU mask is recomputed from the branch inputs (the forward's arithmetic, same bits) and y is not read"""
    _chk(dout, "dout")
    dxs = [torch.empty_like(t) for t in xs]
    ws = workspace("bn", _lib.load().suhpe_bn_workspace_bytes(c), dout.device)
    if betas is not None:
        call("suhpe_repvgg_bn_bwd_remask", _ptr_array(xs), _ptr_array(means), _ptr_array(rstds), _ptr_array(gammas), _ptr_array(betas), len(xs),
             ptr(dout), xs[0].numel() // c, int(rows_a), c, _ptr_array(dxs), _ptr_array(dgammas), _ptr_array(dbetas), ptr(ws))
        return dxs
    _chk(y, "y")
    call("suhpe_repvgg_bn_bwd", _ptr_array(xs), _ptr_array(means), _ptr_array(rstds), _ptr_array(gammas), len(xs), ptr(dout), ptr(y),
         xs[0].numel() // c, int(rows_a), c, _ptr_array(dxs), _ptr_array(dgammas), _ptr_array(dbetas), ptr(ws))
    return dxs


def syncbn_moments(mean, rstd, eps):
    """(E[x], E[x^2]) of this rank's rows as one [2, c] buffer -- what SyncBN averages across the ranks"""
    c = mean.numel()
    buf = torch.empty((2, c), dtype=torch.float32, device=mean.device)
    call("suhpe_syncbn_moments", ptr(mean), ptr(rstd), c, float(eps), ptr(buf))
    return buf


def syncbn_finish(buf, eps, momentum, rows_global, running_mean=None, running_var=None, num_batches_tracked=None):
    """global-batch mean / rstd from the averaged moments + the running statistics' momentum update, one launch"""
    c = buf.shape[1]
    mean = torch.empty(c, dtype=torch.float32, device=buf.device)
    rstd = torch.empty(c, dtype=torch.float32, device=buf.device)
    call("suhpe_syncbn_finish", ptr(buf), c, float(eps), float(momentum), int(rows_global), ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var),
         ptr(num_batches_tracked))
    return mean, rstd


def bn_eval_stats(running_mean, running_var, eps):
    c = running_mean.numel()
    mean = torch.empty(c, dtype=torch.float32, device=running_mean.device)
    rstd = torch.empty(c, dtype=torch.float32, device=running_mean.device)
    call("suhpe_bn_eval_stats", ptr(running_mean), ptr(running_var), c, float(eps), ptr(mean), ptr(rstd))
    return mean, rstd


@_memop
def bn_act_fwd(x, c, mean, rstd, gamma, beta, act, residual=None, post=False, rowscale=None, hw=1, out=None):
    """post=False: act(BN(x) + residual).  post=True: residual + rowscale[sample] * act(BN(x)) (skip + StochasticDepth)."""
    _chk(x, "x"); _chk(residual, "residual"); _chk(rowscale, "rowscale"); _chk(out, "out")
    y = torch.empty_like(x) if out is None else out
    if post:
        call("suhpe_bn_act_skip_fwd", ptr(x), x.numel() // c, c, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act],
             ptr(residual), ptr(rowscale), int(hw), ptr(y))
        return y
    call("suhpe_bn_act_fwd", ptr(x), x.numel() // c, c, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act],
         ptr(residual), ptr(y))
    return y


@_memop
def bn_act_bwd(x, dy, c, mean, rstd, gamma, beta, act, train_stats, dgamma=None, dbeta=None, accumulate=False, y_out=None,
               rowscale=None, hw=1, out=None):
    _chk(x, "x"); _chk(dy, "dy"); _chk(out, "out")
    dx = torch.empty_like(x) if out is None else out
    ws = workspace("bn", _lib.load().suhpe_bn_workspace_bytes(c), x.device)
    if rowscale is not None:
        call("suhpe_bn_act_skip_bwd", ptr(x), ptr(dy), x.numel() // c, c, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ACT[act],
             int(train_stats), ptr(rowscale), int(hw), ptr(dx), ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws))
        return dx
    call("suhpe_bn_act_bwd", ptr(x), ptr(dy), ptr(y_out), x.numel() // c, c, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
         ACT[act], int(train_stats), ptr(dx), ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws))
    return dx


@_memop
def layernorm_fwd(x, c, eps, gamma, beta):
    _chk(x, "x")
    rows = x.numel() // c
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("suhpe_layernorm_fwd", ptr(x), rows, c, float(eps), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd))
    return y, mean, rstd


@_memop
def layernorm_bwd(x, dy, c, mean, rstd, gamma, dgamma=None, dbeta=None, accumulate=False, skip=None):
    """skip: a tensor of x's shape added to dx (the gradient that bypassed the normalised sub-block through its residual connection)"""
    _chk(x, "x"); _chk(dy, "dy"); _chk(skip, "skip")
    dx = torch.empty_like(x)
    assert skip is None or skip.numel() == x.numel()
    ws = workspace("bn", _lib.load().suhpe_bn_workspace_bytes(c), x.device)
    call("suhpe_layernorm_bwd_skip", ptr(x), ptr(dy), x.numel() // c, c, ptr(mean), ptr(rstd), ptr(gamma), ptr(skip), ptr(dx), ptr(dgamma),
         ptr(dbeta), int(accumulate), ptr(ws))
    return dx


@_memop
def act_fwd(x, c, act, bias=None):
    _chk(x, "x")
    y = torch.empty_like(x)
    call("suhpe_act_fwd", ptr(x), x.numel() // c, c, ptr(bias), ACT[act], ptr(y))
    return y


@_memop
def act_bwd(x, dy, c, act, bias=None):
    _chk(x, "x"); _chk(dy, "dy")
    dx = torch.empty_like(x)
    call("suhpe_act_bwd", ptr(x), ptr(dy), x.numel() // c, c, ptr(bias), ACT[act], ptr(dx))
    return dx


# ---------------------------------------------------------------------------------------------- pooling / SE / misc
@_memop
def maxpool_fwd(x, k, stride, pad):
    """nn.MaxPool2d(k, stride, pad) on (N, H, W, C) -> (y, window positions uint8 for the backward)"""
    _chk(x, "x")
    n, h, w, c = x.shape
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    y = torch.empty((n, oh, ow, c), dtype=torch.float32, device=x.device)
    idx = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=x.device)
    call("suhpe_maxpool_fwd", ptr(x), n, h, w, c, int(k), int(stride), int(pad), ptr(y), ptr(idx))
    return y, idx


@_memop
def maxpool_bwd(dy, idx, x_shape, k, stride, pad):
    _chk(dy, "dy")
    n, h, w, c = x_shape
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    call("suhpe_maxpool_bwd", ptr(dy), ptr(idx), n, h, w, c, int(k), int(stride), int(pad), ptr(dx))
    return dx


@_memop
def avgpool_fwd(x):
    """(N,H,W,C) or (N,L,C) -> (N,C)"""
    _chk(x, "x")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    y = torch.empty((n, c), dtype=torch.float32, device=x.device)
    call("suhpe_avgpool_fwd", ptr(x), n, hw, c, ptr(y))
    return y


@_memop
def avgpool_bwd(dy, x_shape):
    _chk(dy, "dy")
    n, c = x_shape[0], x_shape[-1]
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    call("suhpe_avgpool_bwd", ptr(dy), n, dx.numel() // (n * c), c, ptr(dx))
    return dx


@_memop
def se_scale_fwd(x, s):
    _chk(x, "x"); _chk(s, "s")
    n, c = x.shape[0], x.shape[-1]
    y = torch.empty_like(x)
    call("suhpe_se_scale_fwd", ptr(x), ptr(s), n, x.numel() // (n * c), c, ptr(y))
    return y


@_memop
def se_scale_bwd_s(dy, x):
    n, c = x.shape[0], x.shape[-1]
    ds = torch.empty((n, c), dtype=torch.float32, device=x.device)
    call("suhpe_se_scale_bwd_s", ptr(dy), ptr(x), n, x.numel() // (n * c), c, ptr(ds))
    return ds


@_memop
def se_scale_bwd_x(dy, s, dpool):
    n, c = dy.shape[0], dy.shape[-1]
    dx = torch.empty_like(dy)
    call("suhpe_se_scale_bwd_x", ptr(dy), ptr(s), ptr(dpool), n, dy.numel() // (n * c), c, ptr(dx))
    return dx


@_memop
def se_mlp_fwd(p, w1, b1, w2, b2):
    """p (n,c) -> a1 (n,csq), s (n,c)"""
    n, c = p.shape
    csq = w1.shape[0]
    a1 = torch.empty((n, csq), dtype=torch.float32, device=p.device)
    s = torch.empty((n, c), dtype=torch.float32, device=p.device)
    call("suhpe_se_mlp_fwd", ptr(p), ptr(w1), ptr(b1), ptr(w2), ptr(b2), n, c, csq, ptr(a1), ptr(s))
    return a1, s


@_memop
def se_mlp_bwd(ds, s, a1, p, w1, w2, dw1, db1, dw2, db2):
    n, c = p.shape
    csq = w1.shape[0]
    dp = torch.empty((n, c), dtype=torch.float32, device=p.device)
    ws = workspace("se", n * (c + 2 * csq) * 4, p.device)
    call("suhpe_se_mlp_bwd", ptr(ds), ptr(s), ptr(a1), ptr(p), ptr(w1), ptr(w2), n, c, csq, ptr(dp), ptr(dw1), ptr(db1), ptr(dw2),
         ptr(db2), ptr(ws))
    return dp


@_memop
def add_rowscale(a, b, rowscale):
    """a + b * rowscale[n]  (a or rowscale may be None)"""
    _chk(b, "b"); _chk(a, "a")
    n = b.shape[0]
    y = torch.empty_like(b)
    call("suhpe_add_rowscale", ptr(a), ptr(b), ptr(rowscale), n, b.numel() // n, ptr(y))
    return y


@_memop
def add(a, b):
    _chk(a, "a"); _chk(b, "b")
    y = torch.empty_like(a)
    call("suhpe_add", ptr(a), ptr(b), a.numel(), ptr(y))
    return y


def add3(a, b, c):
    """(a + b) + c in one pass"""
    _chk(a, "a"); _chk(b, "b"); _chk(c, "c")
    y = torch.empty_like(a)
    call("suhpe_add3", ptr(a), ptr(b), ptr(c), a.numel(), ptr(y))
    return y


@_memop
def mul(a, b):
    _chk(a, "a"); _chk(b, "b")
    y = torch.empty_like(a)
    call("suhpe_mul", ptr(a), ptr(b), a.numel(), ptr(y))
    return y


_counters = {}
_host_offsets = {}


_reseed_hooks = []


def rng_counter(device, seed=None):
    """Device-resident (seed, offset) pair of the counter-based mask generator; the running position is kept on the
    host (one launch per draw)."""
    key = torch.device(device)
    if key not in _counters or seed is not None:
        s = int(seed if seed is not None else torch.initial_seed()) & ((1 << 63) - 1)
        _counters[key] = torch.tensor([s, 0], dtype=torch.int64).to(key)
        _host_offsets[key] = 0
        for hook in _reseed_hooks:                           # masks drawn ahead under the old seed / position are dropped (layers.prefetch_masks)
            hook()
    return _counters[key]


@_memop
def dropout_mask(shape, keep, device):
    m = torch.empty(shape, dtype=torch.float32, device=device)
    ctr = rng_counter(device)
    key = torch.device(device)
    call("suhpe_dropout_mask", ptr(ctr), _host_offsets[key], 0, float(keep), m.numel(), ptr(m))
    _host_offsets[key] += (m.numel() + 3) // 4
    return m


_MASK_TABLES = {}


def dropout_masks(requests, device):
    """requests: [(shape, keep)] -> list of masks, the values consecutive dropout_mask calls would have drawn, from ONE launch
    (suhpe_dropout_mask_batch); the request table lives on the device and is reused while the sequence repeats"""
    key = (torch.device(device), tuple((tuple(s), float(k)) for s, k in requests))
    tab = _MASK_TABLES.get(key)
    if tab is None:
        counts = [int(torch.Size(s).numel()) for s, _ in requests]
        rel, off, r, o = [], [], 0, 0
        for c in counts:
            rel.append(r); off.append(o)
            r += (c + 3) // 4
            o += (c + 3) // 4 * 4                            # 16-byte aligned sub-buffers
        dev = torch.device(device)
        tab = (torch.tensor(rel, dtype=torch.int64, device=dev), torch.tensor([float(k) for _, k in requests], dtype=torch.float32, device=dev),
               torch.tensor(counts, dtype=torch.int64, device=dev), torch.tensor(off, dtype=torch.int64, device=dev), counts, off, r, o)
        _MASK_TABLES[key] = tab
    rel_t, keep_t, count_t, off_t, counts, off, total4, total = tab
    out = torch.empty(total, dtype=torch.float32, device=device)
    ctr = rng_counter(device)
    dkey = torch.device(device)
    call("suhpe_dropout_mask_batch", ptr(ctr), _host_offsets[dkey], ptr(rel_t), ptr(keep_t), ptr(count_t), ptr(off_t), len(counts), max(counts), ptr(out))
    _host_offsets[dkey] += total4
    return [out[o:o + c].view(tuple(s)) for (s, _), c, o in zip(requests, counts, off)]


def weighted_sum(x, w=None, scale=1.0):
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    call("suhpe_weighted_sum", ptr(x), ptr(w), x.numel(), float(scale), ptr(out))
    return out


def scale_rows(x, g, scale=1.0):
    c = x.shape[-1]
    y = torch.empty_like(x)
    call("suhpe_scale_rows", ptr(x), ptr(g), x.numel() // c, c, float(scale), ptr(y))
    return y


def masked_mean(x, mask):
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    call("suhpe_masked_mean", ptr(x), ptr(mask), x.numel(), ptr(out))
    return out


def broadcast_scale(src, scale, count):
    out = torch.empty(count, dtype=torch.float32, device=src.device)
    call("suhpe_broadcast_scale", ptr(src), float(scale), int(count), ptr(out))
    return out


def axpby(a, alpha, b=None, beta=0.0):
    out = torch.empty_like(a)
    call("suhpe_axpby", ptr(a), float(alpha), ptr(b), float(beta), a.numel(), ptr(out))
    return out


def fill_(t, value):
    call("suhpe_fill", ptr(t), t.numel(), float(value))
    return t


def nchw_to_nhwc(img, out=None):
    _chk(img, "img"); _chk(out, "out")
    n, c, h, w = img.shape
    if out is None:
        out = torch.empty((n, h, w, c), dtype=torch.float32, device=img.device)
    assert tuple(out.shape) == (n, h, w, c)
    call("suhpe_nchw_to_nhwc", ptr(img), n, c, h, w, ptr(out))
    return out


def strong_aug(src, cutout_boxes, cutmix_boxes, perm, holes, src_nhwc=False, dst_nhwc=True, out=None):
    """CutOut + CutMix (+ layout change) out of place.  src (N,C,H,W) or (N,H,W,C)."""
    _chk(src, "src")
    if src_nhwc:
        n, h, w, c = src.shape
    else:
        n, c, h, w = src.shape
    shape = (n, h, w, c) if dst_nhwc else (n, c, h, w)
    _chk(out, "out")
    dst = torch.empty(shape, dtype=torch.float32, device=src.device) if out is None else out
    assert tuple(dst.shape) == shape
    call("suhpe_strong_aug", ptr(src), ptr(dst), n, c, h, w, int(src_nhwc), int(dst_nhwc), ptr(cutout_boxes),
         ptr(cutmix_boxes), ptr(perm), int(holes))
    return dst


# ---------------------------------------------------------------------------------------------- head
def fisher_nll(A, R, overreg):
    _chk(A, "A"); _chk(R, "R")
    n = A.shape[0]
    loss = torch.empty(n, dtype=torch.float32, device=A.device)
    dA = torch.empty((n, 9), dtype=torch.float32, device=A.device)
    Rest = torch.empty((n, 3, 3), dtype=torch.float32, device=A.device)
    call("suhpe_fisher_nll", ptr(A), ptr(R), n, float(overreg), ptr(loss), ptr(dA), ptr(Rest))
    return loss, dA, Rest


def fisher_svd_project(A):
    _chk(A, "A")
    n = A.numel() // 9
    R = torch.empty((n, 3, 3), dtype=torch.float32, device=A.device)
    call("suhpe_fisher_svd_project", ptr(A), n, ptr(R))
    return R


def fisher_entropy(A):
    _chk(A, "A")
    n = A.numel() // 9
    H = torch.empty(n, dtype=torch.float32, device=A.device)
    call("suhpe_fisher_entropy", ptr(A), n, ptr(H))
    return H


def fisher_ce(A1, A2, mask=None):
    _chk(A1, "A1"); _chk(A2, "A2"); _chk(mask, "mask")
    n = A1.numel() // 9
    ce = torch.empty(n, dtype=torch.float32, device=A1.device)
    dA2 = torch.empty((n, 9), dtype=torch.float32, device=A1.device)
    call("suhpe_fisher_ce", ptr(A1), ptr(A2), ptr(mask), n, ptr(ce), ptr(dA2))
    return ce, dA2


def fisher_log_pdf(A, R):
    _chk(A, "A"); _chk(R, "R")
    n = A.numel() // 9
    out = torch.empty(n, dtype=torch.float32, device=A.device)
    call("suhpe_fisher_log_pdf", ptr(A), ptr(R), n, ptr(out))
    return out


def laplace_nll(A, R, grids):
    _chk(A, "A"); _chk(R, "R"); _chk(grids, "grids")
    n = A.numel() // 9
    loss = torch.empty(n, dtype=torch.float32, device=A.device)
    dA = torch.empty((n, 9), dtype=torch.float32, device=A.device)
    mode = torch.empty((n, 3, 3), dtype=torch.float32, device=A.device)
    call("suhpe_laplace_nll", ptr(A), ptr(R), ptr(grids), n, grids.numel() // 9, ptr(loss), ptr(dA), ptr(mode))
    return loss, dA, mode


def rot_adjust(A, aug, mode):
    _chk(A, "A"); _chk(aug, "aug")
    n = A.numel() // 9
    out = torch.empty((n, 9), dtype=torch.float32, device=A.device)
    call("suhpe_rot_adjust", ptr(A), ptr(aug), n, int(mode), ptr(out))
    return out


def euler_from_rot(R, full_range=False):
    _chk(R, "R")
    n = R.numel() // 9
    e = torch.empty((n, 3), dtype=torch.float32, device=R.device)
    call("suhpe_euler_from_rot", ptr(R), n, int(full_range), ptr(e))
    return e


def pose_err_deg(pred, gt, gt_euler=None):
    _chk(pred, "pred"); _chk(gt, "gt"); _chk(gt_euler, "gt_euler")
    n = pred.numel() // 9
    err = torch.empty(n, dtype=torch.float32, device=pred.device)
    call("suhpe_pose_err_deg", ptr(pred), ptr(gt), ptr(gt_euler), n, ptr(err))
    return err


def entropy_mask(H, thres):
    """thres: 1-element device tensor.  -> (mask float (n,), count (1,))"""
    n = H.numel()
    mask = torch.empty(n, dtype=torch.float32, device=H.device)
    count = torch.empty(1, dtype=torch.float32, device=H.device)
    call("suhpe_entropy_mask", ptr(H), ptr(thres), n, ptr(mask), ptr(count))
    return mask, count


# ---------------------------------------------------------------------------------------------- attention
@_memop
def window_attn_fwd(qkv, bias_table, bias_idx, B, H, W, ws, heads, scale, save_probs=True):
    """qkv (B*H*W, heads*96) -> out (B*H*W, heads*32), probs (or None)"""
    _chk(qkv, "qkv"); _chk(bias_table, "bias_table")
    dh = qkv.shape[-1] // (3 * heads)
    out = torch.empty((qkv.shape[0], heads * dh), dtype=torch.float32, device=qkv.device)
    probs = None
    if save_probs:
        n = _lib.load().suhpe_window_attn_probs_elems(B, H, W, ws, heads)
        probs = torch.empty(n, dtype=torch.float32, device=qkv.device)
    ws_buf = workspace("attn_bias", max(4, _lib.load().suhpe_window_attn_fwd_workspace_bytes(ws, heads)), qkv.device)
    call("suhpe_window_attn_fwd", ptr(qkv), ptr(bias_table), ptr(bias_idx), B, H, W, ws, heads, dh, bias_table.shape[1], float(scale),
         ptr(out), ptr(probs), ptr(ws_buf))
    return out, probs


@_memop
def window_attn_bwd(qkv, probs, dout, bias_idx, B, H, W, ws, heads, noff, scale, dbias, inverse=None):
    """inverse = (inv_ptr, inv_list) int32 device tensors (CSR inverse of bias_idx): the bias-table gradient is then reduced in a
    fixed order (bit-reproducible); without it the backward kernel's atomic accumulation is used"""
    _chk(qkv, "qkv"); _chk(dout, "dout")
    dh = qkv.shape[-1] // (3 * heads)
    dqkv = torch.empty_like(qkv)
    ws_buf = workspace("attn", probs.numel() * 4, qkv.device)
    call("suhpe_window_attn_bwd", ptr(qkv), ptr(probs), ptr(dout), ptr(bias_idx), B, H, W, ws, heads, dh, noff, float(scale), ptr(dqkv),
         None if inverse is not None else ptr(dbias), ptr(ws_buf))
    if inverse is not None:
        ws2 = workspace("attn_dbias", _lib.load().suhpe_window_attn_dbias_workspace_bytes(ws, heads), qkv.device)
        call("suhpe_window_attn_dbias", ptr(ws_buf), ptr(inverse[0]), ptr(inverse[1]), B, H, W, ws, heads, noff, ptr(dbias), ptr(ws2))
    return dqkv


@_memop
def window_attn_fwd_rs(qkv, bias_table, bias_idx, B, H, W, ws, heads, scale, save_stats=True):
    """the recomputing form: qkv -> out, rowstats ((windows*heads, NP, 2): row max and 1 / row sum of the softmax; None for inference) -- no N x N
    probabilities leave the kernel"""
    _chk(qkv, "qkv"); _chk(bias_table, "bias_table")
    lib = _lib.load()
    dh = qkv.shape[-1] // (3 * heads)
    out = torch.empty((qkv.shape[0], heads * dh), dtype=torch.float32, device=qkv.device)
    stats = torch.empty(lib.suhpe_window_attn_rowstats_elems(B, H, W, ws, heads), dtype=torch.float32, device=qkv.device) if save_stats else None
    ws_buf = workspace("attn_bias_rs", max(4, lib.suhpe_window_attn_rs_workspace_bytes(ws, heads)), qkv.device)
    call("suhpe_window_attn_fwd_rs", ptr(qkv), ptr(bias_table), ptr(bias_idx), B, H, W, ws, heads, dh, bias_table.shape[1], float(scale),
         ptr(out), ptr(stats), ptr(ws_buf))
    return out, stats


@_memop
def window_attn_bwd_rs(qkv, out, stats, dout, bias_table, bias_idx, B, H, W, ws, heads, noff, scale, dbias, inverse=None):
    """backward of window_attn_fwd_rs from the forward's output and row statistics; `inverse` as in window_attn_bwd"""
    _chk(qkv, "qkv"); _chk(dout, "dout"); _chk(out, "out")
    lib = _lib.load()
    dh = qkv.shape[-1] // (3 * heads)
    dqkv = torch.empty_like(qkv)
    ws_buf = workspace("attn", lib.suhpe_window_attn_probs_elems(B, H, W, ws, heads) * 4, qkv.device)
    ws_bias = workspace("attn_bias_rs", max(4, lib.suhpe_window_attn_rs_workspace_bytes(ws, heads)), qkv.device)
    call("suhpe_window_attn_bwd_rs", ptr(qkv), ptr(out), ptr(stats), ptr(dout), ptr(bias_table), ptr(bias_idx), B, H, W, ws, heads, dh, noff,
         float(scale), ptr(dqkv), None if inverse is not None else ptr(dbias), ptr(ws_buf), ptr(ws_bias))
    if inverse is not None:
        ws2 = workspace("attn_dbias", lib.suhpe_window_attn_dbias_workspace_bytes(ws, heads), qkv.device)
        call("suhpe_window_attn_dbias", ptr(ws_buf), ptr(inverse[0]), ptr(inverse[1]), B, H, W, ws, heads, noff, ptr(dbias), ptr(ws2))
    return dqkv


# ---------------------------------------------------------------------------------------------- optimiser
# Parameter writes that torch's per-tensor _version cannot see (the kernels take raw pointers of the flat buffers): a counter that tapes may record
# to learn whether the weights moved between a forward and its backward (src/repvgg.py: the recomputed ReLU mask of the fused block tail)
PARAM_EPOCH = [0]


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    PARAM_EPOCH[0] += 1
    call("suhpe_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
         float(weight_decay), int(step))


def ema_update(ema, p, alpha):
    PARAM_EPOCH[0] += 1
    call("suhpe_ema_update", ptr(ema), ptr(p), ema.numel(), float(alpha))
