"""ctypes binding of libhvn_hip.so (include/hvn.h).  No fallback: if the library is
missing or no gfx950 device is visible, every entry point raises."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhvn_hip.so")
# Experiment builds of the same sources (`build_variant`), loaded INSTEAD of the default library when HVN_LIB_VARIANT names one:
# kernel A/B runs on one box without rebuilding there.  Not used by the product path (unset = libhvn_hip.so).
VARIANTS = {
    "pad": ("-DHVN_SWZ=0",),                        # padded LDS rows (round-1 layout) instead of the XOR swizzle
    "lin": ("-DHVN_EPI_LINEAR=1",),                 # prepared, NOT yet measured: branch-free epilogue addressing for row-contiguous views
    "nt": ("-DHVN_NT=1",),                          # prepared, NOT yet measured: non-temporal hints on the epilogue's residual loads / stores
    "lin_nt": ("-DHVN_EPI_LINEAR=1", "-DHVN_NT=1"),
    "noxcd": ("-DHVN_WINO_XCD=0", "-DHVN_CONV_XCD_CONTIG=0"),   # A/B: round-robin tile order in the Winograd input transform and the multi-tap convolutions
    "fullepi": ("-DHVN_X3G_FULL_EPI=1",),           # A/B (round 6): hvn_conv_igemm_x3g with the one full epilogue of rounds 1-5 instead of the 8 operand-set forms
    "wn1": ("-DHVN_X3G_WN=1",),                      # A/B (round 6, neutral): hvn_conv_igemm_x3g with 32 x 128 wave tiles (every wave splits its A fragment once)
    "prev": (),                                      # same-box A/B against ANOTHER CHECKOUT's library: built by hand into libhvn_hip_prev.so (never by build_variant)
    "trace": ("-DHVN_TRACE_FINE=1",),               # diagnosis: per-phase timestamps of the conv epilogue (with HVN_CONV_TRACE, tools/conv_trace.py --fine)
}
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ("hvn_conv.hip", "hvn_conv_chain.hip", "hvn_conv_chain_x3.hip", "hvn_conv_chain_x3r.hip", "hvn_conv_bf16.hip", "hvn_conv_bf16g.hip", "hvn_conv_chain_bf16.hip", "hvn_conv_x3.hip", "hvn_conv_x3g.hip", "hvn_net_ops.hip", "hvn_postproc.hip", "hvn_api.hip", "hvn_train.hip", "hvn_wgrad_x3.hip", "hvn_targets.hip", "hvn_wsi_merge.hip",
           "hvn_augment.hip", "hvn_train_api.hip", "hvn_contour.cpp")
HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fvisibility=hidden", "-Wno-unused-value", "-pthread")


class hvn_view(ctypes.Structure):
    _fields_ = [("base", ctypes.c_void_p), ("sn", ctypes.c_int64), ("sy", ctypes.c_int64), ("sx", ctypes.c_int64),
                ("h", ctypes.c_int32), ("w", ctypes.c_int32), ("c", ctypes.c_int32), ("sc", ctypes.c_int32)]


class hvn_op(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ("kind", "kh", "kw", "stride", "pad_t", "pad_l", "relu", "cout", "tile_n", "x_dtype", "groups", "_rsv")] + \
               [("x", hvn_view), ("res", hvn_view), ("y", hvn_view), ("x2", hvn_view)] + \
               [(k, ctypes.c_void_p) for k in ("w", "bias", "pre_scale", "pre_shift", "post_scale", "post_shift")] + \
               [("batch_stride", ctypes.c_int64 * 3), ("nbatch", ctypes.c_int32), ("act_dtype", ctypes.c_int32)] + \
               [("y2", hvn_view), ("w2", ctypes.c_void_p), ("bias2", ctypes.c_void_p), ("cout2", ctypes.c_int32), ("_rsv2", ctypes.c_int32)]


class hvn_top(ctypes.Structure):
    """One launch of the training step (include/hvn.h, training section)."""
    _fields_ = [(k, ctypes.c_int32) for k in ("kind", "kh", "kw", "stride", "pad_t", "pad_l", "groups", "cout", "cin_g", "mode", "lead_pad", "_pad")] + \
               [("x", hvn_view), ("y", hvn_view), ("dx", hvn_view), ("dy", hvn_view), ("p", ctypes.c_void_p * 6),
                ("eps", ctypes.c_float), ("momentum", ctypes.c_float), ("net", ctypes.POINTER(hvn_op)),
                ("batch_stride", ctypes.c_int64 * 3), ("nbatch", ctypes.c_int32), ("_pad2", ctypes.c_int32)]


class hvn_pack_desc(ctypes.Structure):
    """One entry of HVN_T_PACK_MULTI's device table (include/hvn.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p)] + [(k, ctypes.c_int32) for k in ("cout", "cin_g", "groups", "taps", "mode", "lead_pad")] + \
               [("gmat", ctypes.c_void_p)]


class hvn_loss(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("logits_np", "logits_hv", "logits_tp", "true_np", "true_tp", "true_hv",
                                               "grad_np", "grad_hv", "grad_tp", "sums", "sobel_ws")] + \
               [(k, ctypes.c_int32) for k in ("n", "h", "w", "nr_types")] + [("total_pixels", ctypes.c_double), ("weight", ctypes.c_float * 6),
                                                                             ("partials", ctypes.c_void_p), ("partials_cap", ctypes.c_int64)]


class hvn_inst_rec(ctypes.Structure):
    _fields_ = [("label", ctypes.c_int32), ("area", ctypes.c_int32), ("rmin", ctypes.c_int32), ("rmax", ctypes.c_int32),
                ("cmin", ctypes.c_int32), ("cmax", ctypes.c_int32), ("sum_x", ctypes.c_double), ("sum_y", ctypes.c_double),
                ("type", ctypes.c_int32), ("type_count", ctypes.c_int32)]


EXPORTS = (
    "hvn_version", "hvn_build_id", "hvn_last_error", "hvn_device_ok", "hvn_run_plan", "hvn_run_op", "hvn_profile_enable",
    "hvn_profile_conv_ms", "hvn_profile_conv_launches", "hvn_profile_conv_ms_list", "hvn_postproc_workspace_bytes", "hvn_postproc",
    "hvn_postproc_taps", "hvn_postproc_stats", "hvn_instance_table_workspace_bytes", "hvn_instance_table", "hvn_trace_contours",
    "hvn_run_train_plan", "hvn_run_train_plan_ws", "hvn_train_workspace_bytes", "hvn_train_last_error", "hvn_loss_partials_count", "hvn_loss_forward", "hvn_loss_backward", "hvn_adam_step",
    "hvn_extract_patches", "hvn_gen_targets", "hvn_gen_targets_workspace_bytes", "hvn_augment_shape", "hvn_augment_input",
    "hvn_wsi_merge_normal", "hvn_wsi_merge_fixing",
)


class HvnError(RuntimeError):
    pass


def source_id():
    """16 hex digits over every source the library is built from (kernels, headers, flags).  Compiled into the library
    (`hvn_build_id()`), so that a binary which does not match the sources next to it is detected at LOAD time -- the `.so` files are
    git-ignored yet travel to the GPU box, and a stale one would otherwise pass for the current kernels."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(SOURCES) + ["hvn_kernels.h"]:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(os.path.dirname(_HERE), "include", "hvn.h"), "rb").read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def _built_id(path):
    """The id compiled into an existing library (read from the file's bytes: dlopen would pin the old mapping in this process)."""
    if not os.path.exists(path):
        return None
    data = open(path, "rb").read()
    i = data.find(b"hvn-build-id:")
    if i < 0:
        return None
    j = data.find(b"\0", i)
    return data[i + 13:j].decode(errors="replace")


_HIPCC_VERSION = None


def _hipcc_version():
    """`hipcc --version` (bytes): part of every cached object's key -- an object built by another ROCm must not be linked."""
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            _HIPCC_VERSION = subprocess.run(["hipcc", "--version"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=60).stdout
        except (OSError, subprocess.SubprocessError):
            _HIPCC_VERSION = b"hipcc-unknown"
    return _HIPCC_VERSION


def _compile(out, extra, verbose):
    """One object per source, compiled in parallel and cached by the hash of what it is built from (the source, both headers, the
    flags) under csrc/.obj/, then linked: editing one kernel file recompiles that file only (~30 s instead of ~150 s for all).
    The .so still carries the id of ALL sources (`hvn_build_id`, stamped into hvn_api.hip's object)."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor

    sid = source_id() + ("" if not extra else "+" + "".join(extra))
    objdir = os.path.join(CSRC, ".obj")
    os.makedirs(objdir, exist_ok=True)
    hdr = open(os.path.join(CSRC, "hvn_kernels.h"), "rb").read() + open(os.path.join(os.path.dirname(_HERE), "include", "hvn.h"), "rb").read()
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra)
    jobs, objs = [], []
    for src in SOURCES:
        defs = ['-DHVN_BUILD_ID="%s"' % sid] if src == "hvn_api.hip" else []
        h = hashlib.sha256(open(os.path.join(CSRC, src), "rb").read() + hdr + " ".join(cflags + defs).encode() + _hipcc_version()).hexdigest()[:16]
        obj = os.path.join(objdir, "%s.%s.o" % (src, h))
        objs.append(obj)
        if not os.path.exists(obj):
            tag = os.path.join(objdir, "%s.flags-%s" % (src, hashlib.sha256(" ".join(extra).encode()).hexdigest()[:8]))
            if os.path.exists(tag):                 # one cached object per (source, flag set): drop the one this flag set built before
                old = open(tag).read().strip()
                if old and os.path.exists(os.path.join(objdir, old)):
                    os.remove(os.path.join(objdir, old))
            open(tag, "w").write(os.path.basename(obj))
            jobs.append(["hipcc", *cflags, *defs, "-c", os.path.join(CSRC, src), "-o", obj])

    def run(cmd):
        # hipcc writes its output in place: compile / link to a private name and rename on success, so that a build killed mid-write
        # (the GPU scripts wrap everything in `timeout`) or two ranks building at once never leave a truncated file under the final name
        if verbose:
            print(" ".join(cmd))
        final = cmd[-1]
        tmp = "%s.tmp.%d" % (final, os.getpid())
        try:
            subprocess.check_call(cmd[:-1] + [tmp])
            os.replace(tmp, final)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", *objs, "-o", out])
    return out


def build(verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU (under a minute).  Rebuilds whenever the id compiled into the existing
    library differs from the id of the sources (not by mtime)."""
    if _built_id(LIB_PATH) == source_id():
        return LIB_PATH
    return _compile(LIB_PATH, (), verbose)


def build_variant(name, verbose=False):
    """`libhvn_hip_<name>.so`: the library compiled with VARIANTS[name]'s extra flags."""
    out = os.path.join(_HERE, "libhvn_hip_%s.so" % name)
    if _built_id(out) == source_id() + "+" + "".join(VARIANTS[name]):
        return out
    return _compile(out, tuple(VARIANTS[name]), verbose)


def lib_path():
    v = os.environ.get("HVN_LIB_VARIANT", "")
    if not v:
        return LIB_PATH
    if v not in VARIANTS:
        raise HvnError("HVN_LIB_VARIANT=%r: unknown variant (known: %s)" % (v, ", ".join(VARIANTS)))
    return os.path.join(_HERE, "libhvn_hip_%s.so" % v)


_LIB = None


def lib():
    """The loaded library (ctypes.CDLL); raises HvnError if it has not been built."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise HvnError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the HoVer-Net hot path)" % os.path.basename(path))
        # torch ships its own libamdhip64 and must load it FIRST: a process that dlopens this library before importing torch ends up
        # with the system HIP runtime for these kernels and torch's for the streams / allocations handed to them (hvn_device_ok then
        # fails on a GPU box: `build(); smoke()` in one process did, round 4).  Loading torch here makes the order deterministic.
        import torch  # noqa: F401
        L = ctypes.CDLL(path)
        L.hvn_build_id.restype = ctypes.c_char_p
        built = L.hvn_build_id().decode()
        if os.path.isdir(CSRC) and not os.environ.get("HVN_LIB_VARIANT") and built != source_id():
            raise HvnError("%s was built from other sources (library %s, sources %s): run `python -c 'import __graft_entry__ as g; g.build()'`"
                           % (os.path.basename(path), built, source_id()))
        L.hvn_last_error.restype = ctypes.c_char_p
        L.hvn_profile_conv_ms.restype = ctypes.c_double
        L.hvn_profile_conv_ms_list.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.hvn_postproc_workspace_bytes.restype = ctypes.c_size_t
        L.hvn_postproc_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.hvn_run_plan.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.hvn_run_op.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.hvn_postproc.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.hvn_postproc_taps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.hvn_postproc_stats.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.hvn_instance_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_size_t, ctypes.c_void_p]
        L.hvn_instance_table_workspace_bytes.restype = ctypes.c_size_t
        L.hvn_instance_table_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.hvn_trace_contours.restype = ctypes.c_long
        L.hvn_trace_contours.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
        L.hvn_extract_patches.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.hvn_gen_targets_workspace_bytes.restype = ctypes.c_size_t
        L.hvn_gen_targets_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.hvn_gen_targets.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.hvn_augment_shape.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.hvn_augment_input.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_void_p]
        L.hvn_wsi_merge_normal.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_int32, ctypes.c_void_p]
        L.hvn_wsi_merge_fixing.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                           ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.hvn_train_last_error.restype = ctypes.c_char_p
        L.hvn_run_train_plan.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.hvn_run_train_plan_ws.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.hvn_train_workspace_bytes.restype = ctypes.c_size_t
        L.hvn_train_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.hvn_loss_partials_count.restype = ctypes.c_int64
        L.hvn_loss_partials_count.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.hvn_loss_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.hvn_loss_backward.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.hvn_adam_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                    ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
        _LIB = L
    return _LIB


def check(rc, what):
    if rc != 0:
        raise HvnError("%s failed (%d): %s" % (what, rc, lib().hvn_last_error().decode()))


def require_gpu():
    if not lib().hvn_device_ok():
        raise HvnError("no gfx950 (MI355X) device is current; the HoVer-Net hot path has no CPU fallback")
