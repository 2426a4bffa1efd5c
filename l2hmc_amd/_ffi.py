"""ctypes binding of libl2hmc_hip.so (the C ABI declared in include/l2hmc.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails this
module raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C l2hmc_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# the in-tree build; L2HMC_LIB names another BUILD OF THE SAME LIBRARY (kernel experiments, sanitizer / LDS-poison passes:
# tools/pytest_with_lib.py) -- never a fallback: whatever is named must exist and pass the ABI checks of lib()
LIB_PATH = os.path.abspath(os.environ["L2HMC_LIB"]) if os.environ.get("L2HMC_LIB") else os.path.join(_HERE, "csrc", "libl2hmc_hip.so")

ENERGY_GAUSS_DIAG, ENERGY_GAUSS_DENSE, ENERGY_GMM, ENERGY_ROUGHWELL, ENERGY_FUNNEL = 1, 2, 3, 4, 5

_fp = C.c_void_p  # device pointers travel as integers

NET_FIELDS = ("W1", "b1", "W2", "b2", "W3", "b3", "W4", "b4",
              "Ws", "bs", "Wt", "bt", "Wq", "bq", "lam_s", "lam_q")


class L2hmcNet(C.Structure):
    _fields_ = [(k, _fp) for k in NET_FIELDS]


class L2hmcEnergy(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_comp", C.c_int32), ("mu", _fp), ("prec", _fp),
                ("logc", _fp), ("eta", C.c_float), ("easy", C.c_int32),
                ("temperature", C.c_float), ("anneal_beta", C.c_float),
                ("den", C.c_float), ("reserved_", C.c_int32)]


class L2hmcTrajectoryArgs(C.Structure):
    _fields_ = [("packed_nets", _fp), ("energy", L2hmcEnergy), ("masks", _fp), ("trig", _fp),
                ("alpha", _fp), ("eps_host", C.c_float),
                ("n_chains", C.c_int64), ("d", C.c_int32), ("H", C.c_int32), ("T", C.c_int32),
                ("step_begin", C.c_int32), ("n_steps", C.c_int32),
                ("x", _fp), ("v", _fp), ("direction", _fp), ("direction_all", C.c_int32),
                ("u", _fp),
                ("x_out", _fp), ("v_out", _fp), ("logjac_out", _fp), ("p_out", _fp),
                ("x_next", _fp), ("variant", C.c_int32), ("n_proposals", C.c_int32), ("x_hist", _fp),
                ("rng_flags", C.c_uint32), ("rng_seed", C.c_uint64), ("rng_proposal0", C.c_uint64),
                ("chain_offset", C.c_int64),
                ("ais_beta", _fp), ("ais_v0", _fp), ("ais_dbeta", C.c_float), ("ais_refreshment", C.c_float),
                ("ais_w", _fp), ("ais_alpha", _fp)]


RNG_V, RNG_DIR, RNG_U = 1, 2, 4


class L2hmcMlp3(C.Structure):
    _fields_ = [(k, _fp) for k in ("W1", "b1", "W2", "b2", "W3", "b3")] + \
               [(k, C.c_int32) for k in ("n_in", "n_h1", "n_h2", "n_out")]


class L2hmcSplitArgs(C.Structure):
    _fields_ = [("xnet", C.POINTER(L2hmcNet)), ("vnet", C.POINTER(L2hmcNet)), ("H", C.c_int32),
                ("aux_encoder", C.POINTER(L2hmcMlp3)), ("decoder", C.POINTER(L2hmcMlp3)), ("aux", _fp),
                ("masks", _fp), ("trig", _fp), ("alpha", _fp), ("eps_host", C.c_float),
                ("n_chains", C.c_int64), ("d", C.c_int32), ("T", C.c_int32), ("step_begin", C.c_int32),
                ("n_steps", C.c_int32), ("x", _fp), ("v", _fp), ("direction", _fp),
                ("direction_all", C.c_int32), ("u", _fp),
                ("x_out", _fp), ("v_out", _fp), ("logjac_out", _fp), ("p_out", _fp), ("x_next", _fp),
                ("workspace", _fp), ("workspace_floats", C.c_int64), ("hmc", C.c_int32),
                ("bce_scale", C.c_float), ("energy", C.POINTER(L2hmcEnergy)), ("reuse", C.c_int32),
                ("energy_cb", _fp), ("energy_cb_user", _fp), ("gemm_mode", C.c_int32),
                ("net_cb", _fp), ("net_cb_user", _fp)]


# include/l2hmc.h L2hmcEnergyCallback: (user, x, ldx, n_chains, d, U_out, grad_out, ldg, stream) -> int
ENERGY_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                              C.c_int64, C.c_void_p)
# L2hmcNetCallback(user, net, ab, ldab, n_chains, d, it, direction, direction_all, stq_out, stream)
NET_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                           C.c_int32, C.c_void_p, C.c_void_p)
# L2hmcNetVjpCallback(user, net, ab, ldab, n_chains, d, it, direction, direction_all, d_stq, d_ab, ld_dab, stream)   (ABI 6)
NET_VJP_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                               C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
# L2hmcHvpCallback(user, x, ldx, u, ldu, n_chains, d, hv_out, ldhv, stream)
HVP_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                           C.c_void_p, C.c_int64, C.c_void_p)


class L2hmcTrainArgs(C.Structure):
    _fields_ = [("xnet", C.POINTER(L2hmcNet)), ("vnet", C.POINTER(L2hmcNet)), ("energy", L2hmcEnergy),
                ("masks", _fp), ("trig", _fp), ("alpha", _fp), ("eps_host", C.c_float),
                ("n_chains", C.c_int64), ("d", C.c_int32), ("H", C.c_int32), ("T", C.c_int32),
                ("x", _fp), ("v", _fp), ("direction", _fp), ("direction_all", C.c_int32),
                ("scale", C.c_float), ("inv_n", C.c_float),
                ("Lx", _fp), ("p", _fp), ("v1", _fp), ("grad", _fp), ("workspace", _fp), ("variant", C.c_int32)]


class L2hmcTrainSplitArgs(C.Structure):
    _fields_ = [("xnet", C.POINTER(L2hmcNet)), ("vnet", C.POINTER(L2hmcNet)), ("H", C.c_int32),
                ("aux_encoder", C.POINTER(L2hmcMlp3)), ("decoder", C.POINTER(L2hmcMlp3)), ("aux", _fp),
                ("energy", C.POINTER(L2hmcEnergy)), ("hess", _fp),
                ("masks", _fp), ("trig", _fp), ("alpha", _fp), ("eps_host", C.c_float),
                ("n_chains", C.c_int64), ("d", C.c_int32), ("T", C.c_int32),
                ("x", _fp), ("v", _fp), ("direction", _fp), ("direction_all", C.c_int32),
                ("dist_weight", _fp), ("scale", C.c_float), ("inv_n", C.c_float), ("dLx_in", _fp),
                ("Lx", _fp), ("p", _fp), ("v1", _fp), ("dx0_out", _fp), ("grad", _fp),
                ("workspace", _fp), ("workspace_floats", C.c_int64),
                ("energy_scale", C.c_float), ("ediff_out", _fp), ("no_accept", C.c_int32), ("dLv_in", _fp),
                ("dlogjac_in", _fp), ("Lv_out", _fp), ("logjac_out", _fp), ("gemm_mode", C.c_int32), ("net_mode", C.c_int32),
                ("energy_cb", C.c_void_p), ("hvp_cb", C.c_void_p), ("energy_cb_user", C.c_void_p),
                ("net_cb", C.c_void_p), ("net_vjp_cb", C.c_void_p), ("net_cb_user", C.c_void_p)]          # (ABI 6)


class L2hmcTrainStep(C.Structure):
    _fields_ = [("x_head", _fp), ("n_head", C.c_int64), ("u", _fp), ("x_next", _fp), ("terms", _fp), ("loss", _fp),
                ("theta", _fp), ("m", _fp), ("v", _fp), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("epsilon", C.c_float), ("step", C.c_int64), ("train_alpha", C.c_int32)]


STRUCTS = (L2hmcNet, L2hmcEnergy, L2hmcTrajectoryArgs, L2hmcMlp3, L2hmcSplitArgs, L2hmcTrainArgs, L2hmcTrainSplitArgs,
           L2hmcTrainStep)

# every symbol include/l2hmc.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "l2hmc_abi_version": (C.c_int, []),
    "l2hmc_last_error": (C.c_char_p, []),
    "l2hmc_last_kernel": (C.c_int32, [C.c_char_p, C.c_int32]),
    "l2hmc_bf16_planes": (C.c_int, [_fp, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "l2hmc_struct_bytes": (C.c_int64, [C.c_int32]),
    "l2hmc_packed_nets_floats": (C.c_int64, [C.c_int32, C.c_int32]),
    "l2hmc_pack_nets": (C.c_int, [C.POINTER(L2hmcNet), C.POINTER(L2hmcNet), C.c_int32, C.c_int32,
                                  _fp, _fp]),
    "l2hmc_packed_gaussian_floats": (C.c_int64, [C.c_int32]),
    "l2hmc_pack_gaussian": (C.c_int, [_fp, C.c_int32, _fp, _fp]),
    "l2hmc_trajectory": (C.c_int, [C.POINTER(L2hmcTrajectoryArgs), _fp]),
    "l2hmc_energy": (C.c_int, [C.POINTER(L2hmcEnergy), _fp, C.c_int64, C.c_int32, _fp, _fp, _fp]),
    "l2hmc_p_accept": (C.c_int, [C.POINTER(L2hmcEnergy), _fp, _fp, _fp, _fp, _fp, C.c_int64,
                                 C.c_int32, _fp, _fp]),
    "l2hmc_mh_select": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, C.c_int32, _fp, _fp]),
    "l2hmc_loss_terms": (C.c_int, [_fp, C.c_int64, C.c_float, C.c_double, _fp, _fp]),
    "l2hmc_split_workspace_floats": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                                 C.POINTER(L2hmcMlp3), C.POINTER(L2hmcMlp3)]),
    "l2hmc_trajectory_split": (C.c_int, [C.POINTER(L2hmcSplitArgs), _fp]),
    "l2hmc_vae_energy": (C.c_int, [C.POINTER(L2hmcMlp3), _fp, _fp, C.c_int64, C.c_int32, _fp, _fp, _fp, C.c_float, _fp]),
    "l2hmc_p_accept_energies": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_int32, _fp, _fp]),
    "l2hmc_train_workspace_floats": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "l2hmc_train_grad_floats": (C.c_int64, [C.c_int32, C.c_int32]),
    "l2hmc_train_fused_lds_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "l2hmc_train_propose_grad": (C.c_int, [C.POINTER(L2hmcTrainArgs), _fp]),
    "l2hmc_train_step": (C.c_int, [C.POINTER(L2hmcTrainArgs), C.POINTER(L2hmcTrainStep), _fp]),
    "l2hmc_adam_step_terms": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                        C.c_int64, C.c_int32, _fp, C.c_float, _fp, _fp]),
    "l2hmc_train_split_grad_floats": (C.c_int64, [C.c_int32, C.c_int32, C.POINTER(L2hmcMlp3)]),
    "l2hmc_train_split_workspace_floats": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                                       C.POINTER(L2hmcMlp3), C.POINTER(L2hmcMlp3)]),
    "l2hmc_train_split_grad": (C.c_int, [C.POINTER(L2hmcTrainSplitArgs), _fp]),
    "l2hmc_adam_step": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_int64, C.c_int32, _fp]),
    "l2hmc_ais_begin_step": (C.c_int, [_fp, _fp, _fp, C.c_float, C.c_float, _fp, _fp, C.c_int64, C.c_int32, _fp]),
    "l2hmc_ais_end_step": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int64, C.c_int32, _fp]),
    "l2hmc_rng_fill": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                 _fp, _fp, _fp, _fp]),
    "l2hmc_autocov_workspace_doubles": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32]),
    "l2hmc_autocov": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_int64, _fp, _fp, _fp, _fp]),
    "l2hmc_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "l2hmc_step": (C.c_int, [C.POINTER(L2hmcNet), C.POINTER(L2hmcNet), C.POINTER(L2hmcEnergy), _fp, _fp, _fp, _fp, _fp,
                             _fp, C.c_float, C.c_float, C.c_float, _fp, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                             _fp, _fp]),
}

ABI_VERSION = 6          # L2HMC_ABI_VERSION this binding was written against
_lib = None


def lib():
    """The loaded library (raises loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "l2hmc_amd: %s is missing -- the HIP extension has not been built "
                "(run `make -C l2hmc_amd/csrc`); there is no CPU/eager fallback." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        # the version first: a stale library lacks the newer symbols, and "undefined symbol" says less than "rebuild"
        try:
            handle.l2hmc_abi_version.restype, handle.l2hmc_abi_version.argtypes = SYMBOLS["l2hmc_abi_version"]
            found = handle.l2hmc_abi_version()
        except AttributeError:
            found = None
        if found != ABI_VERSION:
            raise RuntimeError("l2hmc_amd: ABI version mismatch (library %s, binding %d): rebuild with "
                               "`make -C l2hmc_amd/csrc`" % (found, ABI_VERSION))
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                raise RuntimeError("l2hmc_amd: %s lacks the symbol %s of ABI %d -- stale build? rebuild with "
                                   "`make -C l2hmc_amd/csrc`" % (LIB_PATH, name, ABI_VERSION))
            fn.restype, fn.argtypes = res, args
        for which, mirror in enumerate(STRUCTS):        # include/l2hmc.h: L2HMC_STRUCT_* in this order
            if handle.l2hmc_struct_bytes(which) != C.sizeof(mirror):
                raise RuntimeError("l2hmc_amd: %s is %d bytes in the library, %d in the binding -- stale build?"
                                   % (mirror.__name__, handle.l2hmc_struct_bytes(which), C.sizeof(mirror)))
        _lib = handle
    return _lib


def last_kernel():
    """Name of the kernel the last trajectory / training call of this thread launched (as rocprofv3 spells it)."""
    buf = C.create_string_buffer(96)
    lib().l2hmc_last_kernel(buf, 96)
    return buf.value.decode()


def check(rc):
    if rc < 0:
        raise RuntimeError("libl2hmc_hip: %s (code %d)" % (lib().l2hmc_last_error().decode(), rc))
    return rc


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(device):
    import torch
    return torch.cuda.current_stream(device).cuda_stream
