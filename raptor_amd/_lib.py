"""ctypes binding of libraptor_quad.so (the C ABI declared in include/raptor_quad.h).

The library is the product: there is no Python/NumPy/torch fallback.  If the shared object
is missing or no HIP device is present the calls fail loudly (``RaptorQuadError``).
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# RAPTOR_QUAD_LIB: another build of the same ABI (tools/ab_build.sh: same-box A/B timing of two revisions)
LIB_PATH = os.environ.get("RAPTOR_QUAD_LIB") or os.path.join(_PKG, "libraptor_quad.so")

POLICY_INPUT_DIM = 22
POLICY_HIDDEN_DIM = 16
POLICY_OUTPUT_DIM = 4
POLICY_NUM_WEIGHTS = 2084
ACTION_DIM = 4
OBSERVATION_DIM = 26
PARAM_DIM = 26
STATE_DIM = 27

ROLLOUT_FUSED, ROLLOUT_CHAINED = 0, 1
ROLLOUT_AUTORESET = 1
POLICY_FP32, POLICY_BF16_MFMA, POLICY_F16X2_MFMA = 0, 1, 2
ACT_IDENTITY, ACT_RELU, ACT_TANH = 0, 1, 2
COMM_ID_BYTES = 128


class RaptorQuadError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libraptor_quad status {status}: {message}")
        self.status = status


class EnvConfig(C.Structure):
    """``rq_env_config`` (include/raptor_quad.h) — what vector.initialize_environment fills."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("dt", C.c_float), ("gravity", C.c_float), ("episode_step_limit", C.c_uint32),
        ("domain_randomization", C.c_uint32),
        ("dr_scale_min", C.c_float), ("dr_scale_max", C.c_float),
        ("dr_thrust_to_weight_min", C.c_float), ("dr_thrust_to_weight_max", C.c_float),
        ("dr_torque_const_min", C.c_float), ("dr_torque_const_max", C.c_float),
        ("dr_motor_tau_min", C.c_float), ("dr_motor_tau_max", C.c_float),
        ("init_guidance", C.c_float), ("init_max_position", C.c_float), ("init_max_angle", C.c_float),
        ("init_max_linear_velocity", C.c_float), ("init_max_angular_velocity", C.c_float),
        ("disturbance_force_std", C.c_float), ("disturbance_torque_std", C.c_float),
        ("noise_position", C.c_float), ("noise_orientation", C.c_float),
        ("noise_linear_velocity", C.c_float), ("noise_angular_velocity", C.c_float),
        ("reward_scale", C.c_float), ("reward_constant", C.c_float),
        ("reward_termination_penalty", C.c_float),
        ("reward_position", C.c_float), ("reward_orientation", C.c_float),
        ("reward_linear_velocity", C.c_float), ("reward_angular_velocity", C.c_float),
        ("reward_action", C.c_float),
        ("termination_enabled", C.c_uint32),
        ("termination_position", C.c_float), ("termination_linear_velocity", C.c_float),
        ("termination_angular_velocity", C.c_float),
        ("action_history_raw", C.c_uint32),
    ]

    def __setattr__(self, name, value):
        # a ctypes Structure takes any attribute name; a misspelt field would silently configure nothing
        if name not in self._field_names:
            raise AttributeError(f"rq_env_config has no field '{name}'")
        super().__setattr__(name, value)

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


EnvConfig._field_names = frozenset(n for n, _ in EnvConfig._fields_)


_vp = C.c_void_p
# array arguments are declared void* so that a plain integer address can be passed: numpy's
# ``a.ctypes.data_as(POINTER(c_float))`` costs 3 us per call, ``a.ctypes.data`` 1 us - five arrays cross the
# boundary in one iteration of the reference's loop
_fp = C.c_void_p
_u32p = C.POINTER(C.c_uint32)
_u8p = C.c_void_p

# name -> argtypes (every function returns int unless listed in _RESTYPES)
_SIGNATURES = {
    "rq_abi_version": [],
    "rq_last_error": [],
    "rq_status_string": [C.c_int],
    "rq_device_count": [C.POINTER(C.c_int)],
    "rq_device_create": [C.c_int, C.POINTER(_vp)],
    "rq_device_destroy": [_vp],
    "rq_device_synchronize": [_vp],
    "rq_device_timer_start": [_vp],
    "rq_device_timer_stop": [_vp, _fp],
    "rq_device_launch_floor": [_vp, C.c_uint32, C.c_uint32, _fp],
    "rq_device_set_rollout_timing": [_vp, C.c_int],
    "rq_device_last_rollout_ms": [_vp, _fp],
    "rq_device_last_rollout_waves": [_vp, C.c_void_p, C.c_uint32, _u32p],
    "rq_device_last_rollout_clock": [_vp, _fp],
    "rq_device_set_speculation": [_vp, C.c_int],
    "rq_device_get_speculation": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _u32p],
    "rq_device_set_resident": [_vp, C.c_int],
    "rq_device_get_resident_timing": [_vp, C.POINTER(C.c_uint64)],
    "rq_device_get_resident": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "rq_device_stream": [_vp, C.POINTER(_vp)],
    "rq_rng_create": [_vp, C.POINTER(_vp)],
    "rq_rng_destroy": [_vp],
    "rq_initialize_rng": [_vp, _vp, C.c_uint64],
    "rq_rng_get": [_vp, C.POINTER(C.c_uint64), _u32p],
    "rq_rng_set_epoch": [_vp, C.c_uint32],
    "rq_env_create": [_vp, C.c_uint32, C.c_uint64, C.POINTER(_vp)],
    "rq_env_destroy": [_vp],
    "rq_env_num_envs": [_vp, _u32p],
    "rq_env_leading_dim": [_vp, _u32p],
    "rq_env_default_config": [C.POINTER(EnvConfig)],
    "rq_initialize_environment": [_vp, _vp],
    "rq_env_set_config": [_vp, C.POINTER(EnvConfig)],
    "rq_env_get_config": [_vp, C.POINTER(EnvConfig)],
    "rq_params_create": [_vp, C.POINTER(_vp)],
    "rq_params_destroy": [_vp],
    "rq_params_get": [_vp, _fp],
    "rq_params_set": [_vp, _fp],
    "rq_params_device_ptr": [_vp, C.POINTER(_vp)],
    "rq_state_create": [_vp, C.POINTER(_vp)],
    "rq_state_destroy": [_vp],
    "rq_state_assign": [_vp, _vp],
    "rq_state_get": [_vp, _fp],
    "rq_state_set": [_vp, _fp],
    "rq_state_device_ptr": [_vp, C.POINTER(_vp)],
    "rq_sample_initial_parameters": [_vp, _vp, _vp, _vp],
    "rq_sample_initial_state": [_vp, _vp, _vp, _vp, _vp],
    "rq_observe": [_vp, _vp, _vp, _vp, _fp, _vp],
    "rq_step": [_vp, _vp, _vp, _vp, _fp, _vp, _vp, _fp],
    "rq_env_observation_device_ptr": [_vp, C.POINTER(_vp)],
    "rq_env_action_device_ptr": [_vp, C.POINTER(_vp)],
    "rq_env_get_observation": [_vp, _fp],
    "rq_env_get_action": [_vp, _fp],
    "rq_env_set_action": [_vp, _fp],
    "rq_env_get_rewards": [_vp, _vp, C.c_int],
    "rq_env_get_terminated": [_vp, _vp, C.c_int],
    "rq_env_get_done_codes": [_vp, _vp, C.c_int],
    "rq_env_get_frozen": [_vp, _vp, C.c_int],
    "rq_env_get_episode_index": [_vp, _vp, C.c_int],
    "rq_env_get_returns": [_vp, _vp, C.c_int],
    "rq_env_get_episode_steps": [_vp, _vp, C.c_int],
    "rq_env_get_finished_returns": [_vp, _vp, C.c_int],
    "rq_env_get_finished_lengths": [_vp, _vp, C.c_int],
    "rq_env_get_finished_counts": [_vp, _vp, C.c_int],
    "rq_env_get_finished_terminated": [_vp, _vp, C.c_int],
    "rq_env_reset_statistics": [_vp],
    "rq_policy_create": [_vp, _fp, C.c_size_t, C.POINTER(_vp)],
    "rq_policy_destroy": [_vp],
    "rq_policy_set_precision": [_vp, C.c_int],
    "rq_policy_pack_image": [_fp, C.c_size_t, C.c_int, _fp, C.c_size_t, C.POINTER(C.c_size_t)],
    "rq_policy_set_standardize": [_vp, _fp, _fp],
    "rq_policy_set_squash": [_vp, C.c_int],
    "rq_policy_set_sample_and_squash": [_vp, C.c_int, _fp, _fp, C.c_uint64],
    "rq_policy_reset": [_vp],
    "rq_policy_evaluate_step": [_vp, _vp, _fp, C.c_uint32, C.c_uint32, _fp],
    "rq_policy_get_hidden": [_vp, _fp, C.c_uint32],
    "rq_policy_set_hidden": [_vp, _fp, C.c_uint32],
    "rq_policy_selftest": [_vp, _fp, _fp, C.c_uint32, C.c_uint32, C.c_float, _fp],
    "rq_policy_evaluate_sequence": [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_int],
    "rq_rollout": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, C.c_int, C.c_uint32],
    "rq_trajectory_create": [_vp, C.c_uint32, C.POINTER(_vp)],
    "rq_trajectory_destroy": [_vp],
    "rq_trajectory_reset": [_vp],
    "rq_trajectory_length": [_vp, _u32p, _u32p],
    "rq_trajectory_get": [_vp, _fp, _fp, _fp, _u8p],
    "rq_trajectory_device_ptrs": [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _u32p],
    "rq_rollout_record": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, C.c_int, C.c_uint32, _vp],
    "rq_trajectory_relabel": [_vp, _vp, _fp, C.c_int],
    "rq_comm_unique_id": [_vp, C.c_size_t],
    "rq_comm_create": [_vp, C.c_uint32, C.c_uint32, _vp, C.c_size_t, C.POINTER(_vp)],
    "rq_comm_destroy": [_vp],
    "rq_comm_info": [_vp, _u32p, _u32p],
    "rq_comm_describe": [_vp, _vp],
    "rq_allgather_returns": [_vp, _vp],
    "rq_comm_gathered": [_vp, C.POINTER(_vp), _u32p, _fp],
    "rq_teacher_bank_create": [_vp, _fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(_vp)],
    "rq_teacher_bank_create_layers": [_vp, _fp, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, C.c_int, C.c_int, C.POINTER(_vp)],
    "rq_teacher_bank_destroy": [_vp],
    "rq_teacher_bank_set_precision": [_vp, C.c_int],
    "rq_trajectory_relabel_teachers": [_vp, _vp, _vp, _fp, C.c_int],
}
_RESTYPES = {"rq_last_error": C.c_char_p, "rq_status_string": C.c_char_p}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None
ABI_VERSION = 5


class _EnvConfigAbi1(C.Structure):
    """rq_env_config as ABI 1 (round 2) declared it: without the trailing action_history_raw"""
    _fields_ = EnvConfig._fields_[:-1]
    _field_names = frozenset(n for n, _ in EnvConfig._fields_[:-1])
    __setattr__ = EnvConfig.__setattr__
    as_dict = EnvConfig.as_dict


_CONFIG_BY_ABI = {1: _EnvConfigAbi1, 2: EnvConfig, 3: EnvConfig, 4: EnvConfig}
_config_type = EnvConfig


def env_config_type():
    """The rq_env_config struct of the LOADED library: EnvConfig, or - under RAPTOR_QUAD_ABI_ANY with an older build -
    that ABI's struct; an ABI whose struct this package does not know is refused rather than handed a struct of the
    wrong size."""
    load()
    if _config_type is None:
        raise RaptorQuadError(-1, f"rq_env_config of ABI {_lib.rq_abi_version()} is unknown to this package (ABI "
                                  f"{ABI_VERSION}): the environment configuration cannot be read or written")
    return _config_type


def _share_hip_runtime_with_torch():
    """One process, one HIP runtime.  A PyTorch-ROCm wheel bundles its own libamdhip64 / libhsa-runtime64;
    if libraptor_quad.so pulls in the system copies first, a later ``import torch`` maps the bundled ones
    beside them and finds no GPU (measured on the MI355X box).  Mapping torch's copies first - by path,
    without importing torch - makes this library bind to them by soname, whichever is imported first.
    Without an installed torch nothing happens and the system ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """Load libraptor_quad.so (built in-tree by ``python -m raptor_amd.build``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RaptorQuadError(-2, f"{LIB_PATH} is missing: run `python -m raptor_amd.build` "
                                      "(there is no fallback implementation)")
        _share_hip_runtime_with_torch()
        lib = C.CDLL(LIB_PATH)
        # RAPTOR_QUAD_ABI_ANY: same-box timing of an OLDER build (tools/ab_run.sh).  Only calls both versions share work:
        # entry points the older library lacks are left unbound (calling one raises AttributeError at the call); the
        # rq_env_*_config calls, whose struct differs between ABI versions, are bound to THAT version's struct below, or
        # refused when this package does not know it (env_config_type)
        any_abi = bool(os.environ.get("RAPTOR_QUAD_ABI_ANY"))
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is None:
                if any_abi:
                    continue
                raise RaptorQuadError(-1, f"{LIB_PATH} does not export {name}: rebuild it (python -m raptor_amd.build)")
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, C.c_int)
        abi = lib.rq_abi_version()
        if abi != ABI_VERSION:
            if not any_abi:
                raise RaptorQuadError(-1, f"ABI version mismatch: raptor_amd speaks {ABI_VERSION}, {LIB_PATH} is {abi}")
            global _config_type
            _config_type = _CONFIG_BY_ABI.get(abi)      # None: the rq_env_*_config calls are refused (env_config_type)
            if _config_type is not None and _config_type is not EnvConfig:
                # ctypes checks pointer types by class: a pointer to the older struct is not a POINTER(EnvConfig)
                ptr = C.POINTER(_config_type)
                for name, argtypes in (("rq_env_default_config", [ptr]), ("rq_env_set_config", [_vp, ptr]), ("rq_env_get_config", [_vp, ptr])):
                    fn = getattr(lib, name, None)
                    if fn is not None:
                        fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(status):
    if status != 0:
        msg = load().rq_last_error()
        raise RaptorQuadError(status, msg.decode() if msg else "")
    return status


_fns = {}


def call(name, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    status = fn(*args)
    if status != 0:
        check(status)
    return status


_from_buffer, _addressof = C.c_char.from_buffer, C.addressof


def _fptr_portable(a):
    try:
        return _addressof(_from_buffer(a))
    except (TypeError, ValueError):
        return a.ctypes.data


try:                                  # the optional CPython helper (csrc/rq_pyfast.c, built by raptor_amd.build): 0.06 us, any array
    from . import _rq_fast as fast
    _fast_address = fast.address
    if not hasattr(fast, "observe") or os.environ.get("RQ_NO_PYFAST"):      # an older build of the helper: addresses only
        fast = None
except ImportError:                   # not built (no gcc / Python.h): the portable ways below
    fast = None
    _fast_address = None

_fn_addrs = {}


def fn_addr(name):
    """Address of a C entry point of the loaded library, for the helper's direct calls (`fast.observe` ...): an int."""
    a = _fn_addrs.get(name)
    if a is None:
        a = _fn_addrs[name] = C.cast(getattr(load(), name), C.c_void_p).value
    return a


def fptr(a):
    """numpy array -> the address of its first element (an int; the argtypes of array arguments are void*).  `a.ctypes.data` builds a
    helper object per call (0.9 us: with four arrays per iteration that was a fifth of the README loop at 8 envs).  With the
    _rq_fast helper `fptr` IS its `address` (one PyObject_GetBuffer, 0.07 us, strided views included); without it: the buffer
    protocol through ctypes (0.4 us; read-only, strided and empty arrays fall back to `.ctypes.data`)."""
    return _fptr_portable(a)


if _fast_address is not None:
    fptr = _fast_address              # noqa: F811  (no Python frame in between)


_ptr_cache = {}


def fptr_cached(a):
    """The same for an array that comes back call after call (the observation buffer of the README loop) when the helper is missing:
    address remembered per object (a weak reference guards against an id being reused) - 0.2 us."""
    e = _ptr_cache.get(id(a))
    if e is not None and e[0]() is a:
        return e[1]
    import weakref
    p = _fptr_portable(a)
    if len(_ptr_cache) > 64:
        _ptr_cache.clear()
    try:
        _ptr_cache[id(a)] = (weakref.ref(a), p)
    except TypeError:
        pass
    return p


if _fast_address is not None:
    fptr_cached = _fast_address       # noqa: F811
