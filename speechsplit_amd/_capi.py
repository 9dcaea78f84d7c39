"""ctypes binding of include/speechsplit_amd.h (the C ABI of libspeechsplit_hip.so).

The product path has no CPU fallback: if the HIP library is missing or fails to load, ``lib()`` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SS_DIAG_LIB=1 (tools/ only): the -DSS_DIAG build with the wrong-result timing experiments compiled in
LIB_PATH = os.environ.get('SS_LIB_PATH') or os.path.join(_HERE, 'lib', 'libspeechsplit_hip_diag.so' if os.environ.get('SS_DIAG_LIB') == '1' else 'libspeechsplit_hip.so')

HP_FIELDS = ('freq', 'dim_neck', 'freq_2', 'dim_neck_2', 'freq_3', 'dim_neck_3', 'dim_enc', 'dim_enc_2', 'dim_enc_3',
             'dim_freq', 'dim_spk_emb', 'dim_f0', 'chs_grp', 'min_len_seg', 'max_len_seg', 'max_len_seq', 'max_len_pad')


class SSHparams(C.Structure):
    _fields_ = [(n, C.c_int) for n in HP_FIELDS]


def hparams_struct(hp):
    """Build the C struct from any attribute bag with the reference's hparams names (hparams.py:9-32)."""
    return SSHparams(**{n: int(getattr(hp, n)) for n in HP_FIELDS})


# every symbol include/speechsplit_amd.h declares: name -> (restype, argtypes)
_vp, _fp, _ip, _i, _l, _f, _d = C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double
SYMBOLS = {
    'ss_last_error': (C.c_char_p, []),
    'ss_abi_version': (_i, []),
    'ss_create': (_vp, [_i, C.POINTER(SSHparams), _i, _i]),
    'ss_destroy': (None, [_vp]),
    'ss_num_params': (_i, [_vp]),
    'ss_param_info': (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_l), C.POINTER(_i), C.POINTER(_l * 3)]),
    'ss_arena_numel': (_l, [_vp]),
    'ss_workspace_bytes': (_l, [_vp]),
    'ss_bind': (_i, [_vp, _fp, _fp, _fp, _fp, _vp, _l, _vp]),
    'ss_g3_forward': (_i, [_vp, _fp, _fp, _fp, _fp, _ip, _i, _i, _i, _fp, _vp]),
    'ss_g3_backward': (_i, [_vp, _fp, _vp]),
    'ss_g3_rhythm': (_i, [_vp, _fp, _i, _i, _fp, _vp]),
    'ss_g6_forward': (_i, [_vp, _fp, _fp, _fp, _ip, _i, _i, _i, _fp, _vp]),
    'ss_g6_backward': (_i, [_vp, _fp, _vp]),
    'ss_g3_train_step': (_i, [_vp, _fp, _fp, _fp, _ip, _fp, _ip, _i, _i, _f, _i, _fp, _vp]),
    'ss_g6_train_step': (_i, [_vp, _fp, _fp, _ip, _fp, _ip, _i, _i, _f, _i, _fp, _vp]),
    'ss_train_finish': (_i, [_vp, _f, _i, _vp]),
    'ss_wait_decoder_grads': (_i, [_vp, _vp]),
    'ss_side_stream': (_vp, [_vp]),
    'ss_grad_split': (_l, [_vp]),
    'ss_stream_report': (C.c_char_p, [_vp]),
    'ss_comm_unique_id': (_i, [C.c_char_p]),
    'ss_comm_init': (_i, [_vp, C.c_char_p, _i, _i]),
    'ss_comm_destroy': (_i, [_vp]),
    'ss_allreduce_grads': (_i, [_vp, _l, _l, _vp]),
    'ss_g3_dp_train_step': (_i, [_vp, _fp, _fp, _fp, _ip, _fp, _ip, _i, _i, _i, _fp, _vp]),
    'ss_g6_dp_train_step': (_i, [_vp, _fp, _fp, _ip, _fp, _ip, _i, _i, _i, _fp, _vp]),
    'ss_set_adam': (_i, [_vp, _d, _d, _d, _d, _l, _vp]),
    'ss_adam_step': (_i, [_vp, _f, _vp]),
    'ss_zero_grads': (_i, [_vp, _vp]),
    'ss_interp_forward': (_i, [_vp, _fp, _ip, _fp, _ip, _i, _i, _i, _fp, _ip, _fp, _ip, _vp]),
    'ss_interp_backward': (_i, [_vp, _fp, _i, _i, _i, _fp, _vp]),
    'ss_check': (_i, [_vp, _vp]),
    'ss_status': (C.c_uint, [_vp]),
    'ss_clear_abort': (_i, [_vp, _vp]),
    'ss_set_lockstep': (_i, [_vp, _i]),
    'ss_op_gemm': (_i, [_fp, _l, _fp, _l, _fp, _l, _fp, _i, _i, _i, _i, _i, _vp]),
    'ss_op_lstm_fwd': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _l, _i, _i, _i, _vp]),
    'ss_op_lstm_bwd': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _l, _i, _i, _i, _vp]),
    'ss_op_lstm_wgrad': (_i, [_fp, _fp, _l, _fp, _fp, _fp, _fp, _fp, _l, _l, _i, _i, _vp]),
    'ss_op_split_image': (_i, [_fp, _l, _l, _i, _f, _fp, _l, _vp]),
    'ss_op_gemm_img': (_i, [_fp, _l, _fp, _l, _fp, _l, _fp, _i, _i, _i, _i, _i, _i, _f, _f, _i, _l, _fp, _vp, _vp]),
    'ss_op_conv_block_scratch': (_l, [_i, _i, _i, _i]),
    'ss_op_conv_block': (_i, [_fp] * 13 + [_l, _i, _i, _i, _i, _vp]),
    'ss_debug_relu_mask': (_i, [_vp, C.c_char_p, _fp, _vp]),
    'ss_melspec_frames': (_i, [_i]),
    'ss_melspec': (_i, [_vp, _i, _vp, _i, _fp, _vp]),
    'ss_f0_normalize': (_i, [_vp, _i, _fp, _vp]),
    'ss_collate': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'ss_set_precision': (_i, [_vp, _i]),
    'ss_profile': (_i, [_vp, C.c_uint]),
    'ss_profile_sample': (_i, [_vp, _i]),
    'ss_scratch_fallbacks': (_l, [_vp]),
    'ss_dp_profile': (_i, [_vp, _i]),
    'ss_dp_profile_read': (_i, [_vp, C.POINTER(C.c_double), _i]),
    'ss_profile_read': (_i, [_vp, _i, C.POINTER(_i), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'ss_profile_timeline': (_i, [_vp, C.POINTER(C.c_double), _i]),
    'ss_debug_gemm_phases': (_i, [C.POINTER(C.c_ulonglong), _i]),
    'ss_debug_xcc_map': (_i, [_i, _i, _i, _i, _vp, _vp]),
    'ss_debug_img_wq': (_i, [C.POINTER(C.c_uint), _i]),
    'ss_tune': (_i, [C.c_char_p, _i]),
    'ss_debug_buffer': (_i, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_l), C.POINTER(_l)]),
    'ss_debug_names': (_i, [_vp, C.c_char_p, _i]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or `make -C speechsplit_amd/csrc`).  There is no CPU fallback for the engine.')
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)          # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = h
        # SS_TUNE="key=value,key=value": experiment knobs applied once at load (A/B runs of the tests and tools against a non-default schedule)
        for kv in filter(None, os.environ.get('SS_TUNE', '').split(',')):
            k, v = kv.split('=')
            if h.ss_tune(k.strip().encode(), int(v)) != 0:
                raise RuntimeError('speechsplit_amd: SS_TUNE: ' + h.ss_last_error().decode())
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError('speechsplit_amd: ' + lib().ss_last_error().decode())
