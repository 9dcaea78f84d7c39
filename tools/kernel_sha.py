"""sha1 over the GEMM kernel sources.  tools/pmc_summary.py stores it with the HBM counters it reads; bench.py drops `roofline.traffic` to
null when the kernels have changed since (a stale profile figure would no longer describe the kernel being timed)."""
import hashlib
import os


def kernel_sources_sha():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'speechsplit_amd', 'csrc')
    h = hashlib.sha1()
    for f in ('gemm_bf16x3.hip', 'gemm_img.hip', 'gemm_f32.hip', 'common.h'):
        h.update(open(os.path.join(root, f), 'rb').read())
    return h.hexdigest()[:16]
