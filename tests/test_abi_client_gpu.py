"""GPU: the drop-in boundary used without torch -- tests/abi_client/abi_client.cpp (plain HIP allocations, include/a2v.h,
liba2v_hip.so) checks a2v_layernorm_fwd, a2v_conv_gemm (Linear + bias + GELU) and a2v_attention_fwd (ALiBi) against CPU
references in double precision."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_cpp_client_of_the_c_abi():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "abi_client", "abi_client.bin")
    if not os.path.exists(exe):
        from animal2vec_amd import build
        build.build_abi_client(force=True, verbose=False)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ABI_CLIENT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
