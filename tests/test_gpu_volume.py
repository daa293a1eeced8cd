"""sdirt_amd.volume.VolumeStepper -- the render loop of a PSF volume with one library call per step (sdirt_psf_call on a
batch with one workgroup per point, the trip rule evaluated on the device) -- against the general call Lensgroup.psf_lr:
same draws, same kernels, same rule, so the same PSFs bit for bit wherever the tiles are float64 (ks <= 49)."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_state, make_lens

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DP = (0.78, 1.44, 0.3, 0.5)


def grid(n, seed=3):
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(n, 3, generator=g) * 2 - 1
    p[:, 2] = -(200 + 19800 * torch.rand(n, generator=g))
    return p


@pytest.mark.parametrize("ks,spp,n", [(21, 4096, 1500), (33, 2048, 1100), (65, 4096, 1200)])
def test_stepper_equals_psf_lr_step_after_step(ks, spp, n):
    from sdirt_amd.volume import VolumeStepper
    lens = make_lens("rf50mm", DEV)
    pts = grid(n).to(DEV)
    torch.manual_seed(11)
    want = [tuple(t.clone() for t in lens.psf_lr(pts, ks=ks, spp=spp, dp=DP)) for _ in range(3)]
    r0 = lens.trips.relaunches
    lens2 = make_lens("rf50mm", DEV)
    torch.manual_seed(11)
    st = VolumeStepper(lens2, pts, ks=ks, spp=spp, dp=DP, depth=2)
    outs = [st.step().clone() if False else st.step() for _ in range(3)]      # three different slots (depth 2)
    st.fence()
    # (random points, fresh pupil samples every step: the batch-wide tables may flip between neighbours from step to step
    # -- both routes then render such a step twice)
    print(f"relaunches: psf_lr {r0}, stepper {st.relaunches}")
    for i, (blk, (L, R)) in enumerate(zip(outs, want)):
        dl, dr = (blk[:, 0] - L).abs().max().item(), (blk[:, 1] - R).abs().max().item()
        if 2 * ks * ks * 8 <= 39 * 1024:          # float64 tiles: run-to-run identical sums
            assert dl == 0.0 and dr == 0.0, (i, dl, dr)
        else:                                     # float tiles: LDS-atomic arrival order in the last bits
            assert dl < 4e-6 and dr < 4e-6, (i, dl, dr)          # measured 1-2e-6 between two renders of one batch
    # the generator is where three psf_lr calls leave it
    torch.manual_seed(11)
    for _ in range(3):
        torch.rand(2 * spp + 2 * 2048)
    nxt = torch.rand(4)
    torch.manual_seed(11)
    lens3 = make_lens("rf50mm", DEV)
    st3 = VolumeStepper(lens3, pts, ks=ks, spp=spp, dp=DP)
    for _ in range(3):
        st3.step()
    st3.fence()
    assert torch.equal(torch.rand(4), nxt)


def test_stepper_corrects_a_wrong_bet_like_the_general_call():
    """Tables that are not the reference's for the batch (one trip short on one surface, one too many on another):
    the status word comes back non-zero, the step is rendered again with the tables the device derived, later steps
    speculate those -- and every block equals psf_lr's."""
    from sdirt_amd.volume import VolumeStepper
    lens = make_lens("rf50mm", DEV)
    pts = grid(1300, seed=9).to(DEV)
    ks, spp = 21, 4096
    torch.manual_seed(5)
    want = [tuple(t.clone() for t in lens.psf_lr(pts, ks=ks, spp=spp, dp=DP)) for _ in range(3)]
    lens2 = make_lens("rf50mm", DEV)
    torch.manual_seed(5)
    st = VolumeStepper(lens2, pts, ks=ks, spp=spp, dp=DP, depth=2)      # three slots: every block is kept
    right = [t.copy() for t in st.tables]
    bad_p, bad_c = right[0].copy(), right[1].copy()
    bad_p[2] -= 1
    bad_c[8] += 1
    st._tp = (C.c_int32 * st.K)(*[int(v) for v in bad_p])
    st._tc = (C.c_int32 * st.K)(*[int(v) for v in bad_c])
    for s in st._slots:
        st._bind(s)
    outs = [st.step() for _ in range(3)]
    st.fence()
    assert st.relaunches >= 1
    assert all(np.array_equal(a, b) for a, b in zip(st.tables, right))
    for blk, (L, R) in zip(outs, want):
        assert torch.equal(blk[:, 0], L) and torch.equal(blk[:, 1], R)


def test_psf_call_status_word_and_corrected_tables_for_one_workgroup_per_point():
    """sdirt_psf_call on a batch with one workgroup per point: 10 trips everywhere -> status 1 | 2 | 4 and the
    reference's tables in SDIRT_CTL_TRIPS2; those tables -> status 0 and the masks newton.verify accepts."""
    from sdirt_amd import _lib, newton
    from sdirt_amd.basics import dptr, stream_ptr
    lens = make_lens("rf50mm", DEV)
    st = load_state("rf50mm")
    pts = grid(300, seed=2).to(DEV)
    ks, S, Sc, K = 21, 1024, 2048, len(lens.surfaces)
    torch.manual_seed(3)
    L0, R0 = lens.psf_lr(pts, ks=ks, spp=S, dp=DP)
    ref_p = lens.trips.cache[("psf", 0.589, "lean")]
    ref_c = lens.trips.cache[("center", "lean")]
    h = _lib.lib()
    po = lens._points_to_object(pts)
    torch.manual_seed(3)
    u = torch.rand(2 * S + 2 * Sc).pin_memory()
    nbytes = int(h.sdirt_psf_call_scratch_bytes(300, S, Sc))
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    ctl = torch.zeros(_lib.CTL_WORDS, dtype=torch.int32).pin_memory()
    cen = torch.empty((300, 2), device=DEV)
    out = torch.empty((300, 2, ks, ks), device=DEV)
    dpp = _lib.DpParams(*DP)
    curved = lens._curved()

    def call(tp, tc):
        _lib.check(h.sdirt_psf_call(lens.dev_lens(0.589), lens.dev_lens(0.589), dptr(po), 300, C.c_void_p(u.data_ptr()), S, Sc,
                                    st["pupil_r"], st["pupil_r"] * 0.25, st["pupil_z"], st["d_sensor"], st["pixel_size"], ks,
                                    C.byref(dpp), (C.c_int32 * K)(*tp), (C.c_int32 * K)(*tc),
                                    _lib.PSF_NORMALIZE | _lib.PSF_INTERLEAVED | _lib.PSF_ZERO_CTL, dptr(cen), dptr(out[:, 0]),
                                    C.c_void_p(out.data_ptr() + 4 * ks * ks), dptr(scratch), C.c_void_p(ctl.data_ptr()),
                                    stream_ptr(torch.device(DEV))))
        torch.cuda.synchronize()
        w = ctl.numpy().view(np.uint32).copy()
        unpack = lambda off: [int(np.int8((int(w[off + (k >> 2)]) >> ((k & 3) * 8)) & 0xFF)) for k in range(K)]
        return int(w[_lib.CTL_STATUS]), int(w[_lib.CTL_ANY_VALID]), unpack(_lib.CTL_TRIPS2), unpack(_lib.CTL_TRIPS2 + 16), w
    full = [10 if c else 0 for c in curved]
    status, anyv, tp, tc, _ = call(full, full)
    assert status == 7 and anyv == 1
    assert tp == [int(v) for v in ref_p] and tc == [int(v) for v in ref_c]
    status, anyv, tp2, tc2, w = call(tp, tc)
    assert status == 0 and anyv == 1 and tp2 == tp and tc2 == tc
    order = list(range(K))
    assert newton.verify(tp, w[_lib.CTL_MASKS:_lib.CTL_MASKS + K], order, curved)[0]
    assert newton.verify(tc, w[_lib.CTL_MASKS + 64:_lib.CTL_MASKS + 64 + K], order, curved)[0]
    assert torch.equal(out[:, 0], L0) and torch.equal(out[:, 1], R0)
    # uniforms in PAGEABLE memory (against the contract: not mapped into the device's address space) go through the copy
    keep, u = u, u.clone()
    assert not u.is_pinned()
    out.zero_()
    status, anyv, _, _, _ = call(tp, tc)
    assert status == 0 and torch.equal(out[:, 0], L0) and torch.equal(out[:, 1], R0)
    u = keep


def test_lanes_round_trip_keeps_the_control_block():
    """sdirt_ctl_to_lanes -> (what an all-reduce(MAX) over identical ranks leaves) -> sdirt_ctl_from_lanes: the same
    masks, flag and uniform-sum words, and the same status as the device's own rule."""
    from sdirt_amd import _lib
    from sdirt_amd.basics import dptr, stream_ptr
    lens = make_lens("rf50mm", DEV)
    K = len(lens.surfaces)
    g = torch.Generator().manual_seed(1)
    ctl = torch.zeros(_lib.CTL_WORDS, dtype=torch.int32)
    ctl[_lib.CTL_MASKS:_lib.CTL_MASKS + 128] = torch.randint(0, 2048, (128,), generator=g, dtype=torch.int32)
    ctl[_lib.CTL_ANY_VALID] = 1
    d = ctl.to(DEV)
    lanes = torch.full((_lib.CTL_LANES,), -7, dtype=torch.int32, device=DEV)
    h, sp = _lib.lib(), stream_ptr(torch.device(DEV))
    _lib.check(h.sdirt_ctl_to_lanes(dptr(d), 12345, dptr(lanes), sp))
    lv = lanes.cpu().numpy()
    assert set(np.unique(lv[:2 * 64 * 11 + 1])) <= {0, 1}
    back = torch.zeros_like(d)
    host = torch.zeros(_lib.CTL_WORDS, dtype=torch.int32).pin_memory()
    tp = (C.c_int32 * K)(*[3] * K)
    _lib.check(h.sdirt_ctl_from_lanes(dptr(lanes), lens.dev_lens(0.589), tp, tp, dptr(back), C.c_void_p(host.data_ptr()), sp))
    torch.cuda.synchronize()
    b = back.cpu()
    assert torch.equal(b[_lib.CTL_MASKS:_lib.CTL_MASKS + 128], ctl[_lib.CTL_MASKS:_lib.CTL_MASKS + 128])
    assert int(b[_lib.CTL_ANY_VALID]) == 1 and int(b[2]) == 12345 and int(b[3]) == 0x3FFFFFFF - 12345
    assert torch.equal(host, b) and int(b[_lib.CTL_STATUS]) != 0      # random masks: the rule rejects [3] * K


def test_deterministic_flag_makes_65x65_grids_repeat_from_run_to_run():
    """SDIRT_PSF_DETERMINISTIC (Lensgroup.deterministic = True): BASELINE config 2's 65 x 65 grids summed in float64 tiles
    by 1024-thread workgroups -- three renders of the same batch with the same pupil points are IDENTICAL (the default's
    fp32 tiles differ in the last bits with the LDS-atomic arrival order), within the splat bar of the default path, the
    chief-ray centres within one ulp of its; beyond ks 70 the library says so."""
    from sdirt_amd import _lib
    lens = make_lens("rf50mm", DEV)
    pts = grid(1500, seed=4).to(DEV)
    ks, spp = 65, 4096
    g = torch.Generator().manual_seed(8)
    st = load_state("rf50mm")
    ang, r2 = torch.rand(spp, generator=g) * 2 * np.pi, torch.rand(spp, generator=g)
    xy = (torch.sqrt(r2) * st["pupil_r"] * torch.cos(ang), torch.sqrt(r2) * st["pupil_r"] * torch.sin(ang))
    angc, r2c = torch.rand(2048, generator=g) * 2 * np.pi, torch.rand(2048, generator=g)
    xyc = (torch.sqrt(r2c) * st["pupil_r"] * 0.25 * torch.cos(angc), torch.sqrt(r2c) * st["pupil_r"] * 0.25 * torch.sin(angc))

    def render():
        cen = torch.empty((1500, 2), device=DEV)
        L, R = lens.psf_lr(pts, ks=ks, spp=spp, dp=DP, pupil_xy=xy, center_pupil_xy=xyc, center_out=cen)
        return L.clone(), R.clone(), cen
    plain = [render() for _ in range(2)]
    lens.deterministic = True
    det = [render() for _ in range(3)]
    for L, R, cen in det[1:]:
        assert torch.equal(L, det[0][0]) and torch.equal(R, det[0][1]) and torch.equal(cen, det[0][2])
    dl = (det[0][0] - plain[0][0]).abs().max().item()
    dr = (det[0][1] - plain[0][1]).abs().max().item()
    assert dl < 4e-6 and dr < 4e-6, (dl, dr)                 # fp32-tile sums against float64-tile sums: 1-2e-6 measured
    from conftest import ulp_diff
    assert ulp_diff(det[0][2].cpu().numpy(), plain[0][2].cpu().numpy()).max() <= 1
    repeat = torch.equal(plain[0][0], plain[1][0]) and torch.equal(plain[0][1], plain[1][1])
    print(f"deterministic vs default: {dl:.2e} / {dr:.2e}; the default repeated itself bit for bit this time: {repeat}")
    with pytest.raises(_lib.SdirtError, match="SDIRT_PSF_DETERMINISTIC"):
        lens.psf_lr(pts[:1100], ks=75, spp=1024, dp=DP)
    lens.psf_lr(pts[:1100], ks=45, spp=1024, dp=DP)               # float64 tiles anyway: nothing to refuse


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SDIRT_FUZZ_SEEDS", 4))))
def test_stepper_fuzz_equals_psf_lr_bit_for_bit(seed):
    """Random batch shapes with one workgroup per point (many points, or few samples), random grid sizes with float64
    tiles (ks <= 49), random dual-pixel geometry, either lens: three steps of the loop == three psf_lr calls from the same
    seed, bit for bit, whatever the trip tables do from step to step."""
    from sdirt_amd.volume import VolumeStepper
    rng = np.random.default_rng(1000 + seed)
    name = "rf35mm" if rng.random() < 0.3 else "rf50mm"
    if rng.random() < 0.5:
        n, spp = int(rng.integers(1024, 2600)), int(rng.choice([256, 1000, 2048, 4096]))
    else:
        n, spp = int(rng.integers(1, 900)), int(rng.choice([64, 256, 777, 1024]))
    ks = int(rng.integers(3, 50))
    dp = (float(rng.uniform(0.6, 0.9)), float(rng.uniform(1.2, 1.6)), float(rng.uniform(0.2, 0.4)), float(rng.uniform(0.3, 0.5)))
    pts = grid(n, seed=seed).to(DEV)
    lens = make_lens(name, DEV)
    if lens._spp_slices(n, spp) != 1:
        pytest.skip("the spp axis is cut for this shape")
    torch.manual_seed(seed)
    want = [tuple(t.clone() for t in lens.psf_lr(pts, ks=ks, spp=spp, dp=dp)) for _ in range(3)]
    lens2 = make_lens(name, DEV)
    torch.manual_seed(seed)
    st = VolumeStepper(lens2, pts, ks=ks, spp=spp, dp=dp, depth=2, streams=int(rng.integers(1, 3)))
    outs = [st.step() for _ in range(3)]
    st.fence()
    for i, (blk, (L, R)) in enumerate(zip(outs, want)):
        assert torch.equal(blk[:, 0], L) and torch.equal(blk[:, 1], R), (name, n, spp, ks, i)
