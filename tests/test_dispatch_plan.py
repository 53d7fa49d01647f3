"""CPU-side sweep over the GEMM dispatch rules (round 6, VERDICT r5 item 6 / "dispatch rules are fitted constants"): dk_gemm_plan asks the launchers
themselves what they would launch -- no kernel runs, no GPU is needed; without a device the rules assume the MI355X's 256 compute units -- for the
block Linears of FLUX and SD3 over the resolutions and batch sizes a user can ask for (512 x 512 is what the reference's CLI defaults to,
mlx/scripts/generate_images.py:15-30).  The properties pinned here are the ones whose absence was the 512 x 512 cliff this round found:
a long-K Linear of at most half a round of tiles must be cut along K, the measured 1024 x 1024 choices must stay what the profiles were taken on,
and no launch of a production shape may leave more than two thirds of the chip idle without a K split."""
import ctypes as C

import pytest

from diffusionkit_amd import _lib

FAKE_WS = 0x10000  # (never dereferenced in plan mode: only "non-NULL, 256-byte aligned" matters)


def plan(M, N, K, ws=True, M2=0, lda=None, ldw=0):
    lib = _lib.load()

    def desc(m):
        d = _lib.dk_gemm_desc()
        d.M, d.N, d.K = m, N, K
        d.lda, d.ldc, d.ldr = lda or K, N, N
        d.alpha, d.epilogue, d.ldw = 1.0, 0, ldw
        if ws:
            d.workspace, d.workspace_bytes = FAKE_WS, lib.dk_gemm_workspace_bytes()
        return d
    p = _lib.dk_gemm_plan_t()
    a = desc(M)
    b = desc(M2) if M2 else None
    rc = lib.dk_gemm_plan(C.byref(a), C.byref(b) if b is not None else None, C.byref(p))
    assert rc == 0, lib.dk_last_error()
    return p


def flux_linears(h=3072):
    return {"qkv": (3 * h, h), "o_proj": (h, h), "fc1": (4 * h, h), "fc2": (h, 4 * h), "linear1": (7 * h, h), "linear2": (h, 5 * h)}


def test_plan_runs_without_a_gpu_and_reports_the_mi355x():
    p = plan(4352, 3072, 3072)
    assert p.n_cu == 256 or p.n_cu > 0
    assert p.kernel in (3, 4, 128) and p.launches == 1 and p.tiles > 0


def test_headline_shapes_keep_their_measured_choices():
    """FLUX.1-schnell 1024 x 1024 (M = 4352; double blocks: 4096 image + 256 text rows grouped): the one-wave-per-SIMD kernel on every block Linear, the
    tile heights of profiles/r05_flux_schnell_kernel_stats.md / r06_*: 224-row tiles where they save a round, 256 on linear1 (1428 tiles = 5.6 rounds)"""
    lin = flux_linears()
    p = plan(4352, *lin["linear1"])
    assert (p.kernel, p.tile_rows, p.tiles, p.split_tiles) == (4, 256, 17 * 84, 0)
    p = plan(4352, *lin["linear2"])
    assert (p.kernel, p.tile_rows, p.tiles, p.split_tiles) == (4, 224, 20 * 12, 0)
    for name in ("qkv", "o_proj", "fc1", "fc2"):  # grouped image + text launches of the double blocks
        p = plan(4096, *lin[name], M2=256)
        assert p.kernel == 4 and p.split_tiles == 0, name
        assert p.tile_rows == 224 and p.tiles == 21 * (lin[name][0] // 256), name


@pytest.mark.parametrize("res,batch", [(512, 1), (640, 1), (768, 1), (512, 2)])
def test_small_launches_with_long_reductions_are_cut_along_k(res, batch):
    """below 1024 x 1024 fc2 / linear2 are a fraction of a round of 256 x 256 tiles: every tile is cut along K (gemm256v3.hip's split; the cliff of
    profiles/r06_flux_512_kernel_stats_before.md: 72 workgroups, 229 us), and o_proj (K = 3072) only where four ranges save 36 steps"""
    S_i = (res // 16) ** 2
    lin = flux_linears()
    M = batch * (256 + S_i)
    p = plan(M, *lin["linear2"])
    tiles256 = -(-M // 256) * 12
    if tiles256 * 2 <= 256:
        assert p.kernel == 3 and p.split_tiles == p.tiles and p.k_pieces >= 2, (res, batch, p.kernel, p.tiles, p.split_tiles)
        assert p.workgroups > 128 and p.ks <= lin["linear2"][1] // 64 // 2
        q = plan(batch * S_i, *lin["fc2"], M2=batch * 256)
        assert q.kernel == 3 and q.split_tiles == q.tiles
        # o_proj (K = 3072: 48 steps): cut only where that saves a workgroup at least 32 steps -- four ranges at 512 x 512 (36), not two at 768 x 768 (24)
        o = plan(batch * S_i, *lin["o_proj"], M2=batch * 256)
        S = min(4, 256 // o.tiles)
        assert (o.split_tiles > 0) == (48 - -(-48 // S) >= 32), (res, batch, o.tiles, o.split_tiles)
    # without the workspace nothing can be cut -- and nothing is
    assert plan(M, *lin["linear2"], ws=False).split_tiles == 0


def test_no_production_launch_wastes_the_chip_silently():
    """the sweep the 1024 x 1024-fitted constants never had: FLUX and SD3-medium block Linears over resolutions 384 ... 1536 and batches 1 ... 8.  A launch
    may under-fill the CUs when its reduction is short (K <= 3072: the split's slab round trips cost more than they save), never when it is long."""
    worst = []
    for h, txt, fam in ((3072, 256, "flux"), (1536, 589, "sd3")):
        for res in (384, 512, 640, 768, 896, 1024, 1280, 1536):
            for batch in (1, 2, 4, 8):
                rows = batch * (2 if fam == "sd3" else 1)
                M_img, M_txt = rows * (res // 16) ** 2, rows * txt
                for name, (N, K) in flux_linears(h).items():
                    if fam == "sd3" and name.startswith("linear"):
                        continue
                    single = name.startswith("linear")
                    p = plan(M_img + M_txt, N, K) if single else plan(M_img, N, K, M2=M_txt)
                    assert p.launches >= 1 and p.workgroups >= p.tiles
                    fill = min(1.0, p.workgroups / p.n_cu)
                    if K >= 6144 and p.kernel != 128:
                        worst.append((fill, fam, res, batch, name, p.kernel, p.tiles, p.workgroups))
    worst.sort()
    fill, *what = worst[0]
    assert fill >= 0.55, f"long-K launch fills only {fill:.2f} of the CUs: {what}"


def test_plan_mode_leaves_no_state_behind():
    """two plans of the same shape agree, and a forced kernel choice shows up in the plan (dk_tune_set drives the same code)"""
    lib = _lib.load()
    a, b = plan(1280, 3072, 15360), plan(1280, 3072, 15360)
    assert [getattr(a, f) for f, _ in a._fields_] == [getattr(b, f) for f, _ in b._fields_]
    try:
        assert lib.dk_tune_set(b"gemm_split", 0) == 0
        assert plan(1280, 3072, 15360).split_tiles == 0
    finally:
        lib.dk_tune_set(b"gemm_split", -1)
    assert plan(1280, 3072, 15360).split_tiles > 0
