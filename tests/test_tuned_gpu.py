"""The shipped library-GEMM selections (mmlearn_amd/tuned/gemm_gfx950.csv) on this box: TunableOp accepts the file, and every
"tn" entry -- run at its exact shape -- gives the product the library's default kernel gives, to bf16 rounding."""

import re

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_selected_library_kernels_compute_the_same_products():
    from mmlearn_amd import tuned

    dev = torch.device("cuda", 0)
    entries = []
    for ln in open(tuned.DEFAULT_FILE).read().splitlines():
        if ln.startswith("Validator"):
            continue
        op, key, sol, _ = ln.split(",")
        m = re.match(r"tn_(\d+)_(\d+)_(\d+)_ld_", key)
        if m and sol != "Default":
            entries.append((op.startswith("GemmAndBias"), *[int(v) for v in m.groups()]))
    assert len(entries) >= 10
    torch.manual_seed(0)
    cases = []
    for with_bias, n_out, rows, k in entries:       # F.linear(x [rows, k], w [n_out, k]) is the library's tn_{n_out}_{rows}_{k}
        x = torch.randn(rows, k, device=dev).bfloat16()
        w = (torch.randn(n_out, k, device=dev) / k ** 0.5).bfloat16()
        b = torch.randn(n_out, device=dev).bfloat16() if with_bias else None
        ref = F.linear(x, w, b).float()
        cases.append((x, w, b, ref.abs().max().item(), ref.to(torch.bfloat16)))
        del ref
    try:
        if not tuned.enable():
            # another PyTorch / hipBLASLt / rocBLAS build than the one the file was recorded on: the product then runs the library's
            # defaults (bench.py says so in its line); nothing to check here
            pytest.skip("TunableOp's validators refused the shipped selections on this box")
        import torch.cuda.tunable as tunable

        assert tunable.is_enabled() and not tunable.tuning_is_enabled()
        for x, w, b, scale, ref16 in cases:
            out = F.linear(x, w, b)
            err = (out.float() - ref16.float()).abs().max().item()
            assert err <= 2e-2 * scale, (tuple(x.shape), tuple(w.shape), err, scale)
    finally:
        tuned.disable()


def test_a_selections_file_from_another_library_build_is_refused(tmp_path):
    """TunableOp's validators (PyTorch / hipBLASLt / rocBLAS versions, architecture) guard the solution indices: a file recorded
    elsewhere must leave every GEMM on the library's default instead of picking kernels by a stale index."""
    import torch.cuda.tunable as tunable

    from mmlearn_amd import tuned

    text = open(tuned.DEFAULT_FILE).read().splitlines()
    bad = [ln if not ln.startswith("Validator,HIPBLASLT_VERSION") else "Validator,HIPBLASLT_VERSION,0-other-build" for ln in text]
    p = tmp_path / "other_build.csv"
    p.write_text("\n".join(bad) + "\n")
    try:
        assert tuned.enable(str(p)) is False
        assert not tunable.is_enabled()
        x = torch.randn(512, 768, device="cuda").bfloat16()
        w = torch.randn(768, 768, device="cuda").bfloat16()
        assert torch.isfinite(F.linear(x, w).float()).all()
    finally:
        tuned.disable()
