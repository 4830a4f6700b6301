"""CPU checks of the measurement tools under tools/ (no GPU, no library)."""
import os


def test_decode_timeline_tool_on_a_synthetic_trace(tmp_path):
    """tools/rocpd_timeline.py (the begin/end accounting of DESIGN.md section 5) on a hand-made rocpd-shaped database: two warm-up
    dispatches, then 6 decode tokens of a 2-layer model (5 dispatches per layer + LM head + finish), 0.5 us between dispatches."""
    import sqlite3
    import subprocess
    import sys
    db = tmp_path / "trace.db"
    con = sqlite3.connect(db)
    con.execute("create table kernels(name text, start int, end int)")
    t = [1000]

    def k(name, dur):
        t[0] += 500
        con.execute("insert into kernels values(?,?,?)", (name, t[0], t[0] + dur))
        t[0] += dur

    k("void jh::embed_rows_kernel(void const*)", 4000)
    k("void jh::finish_token_kernel(float const*)", 3000)
    for _ in range(6):
        for _ in range(2):
            k("void jh::gemv_i8q4_kernel<1, 0, 2, 2, 0>(jh::GemvParams)", 5800)
            k("void jh::attn_decode_kernel<128, 4, 2>(jh::AttnParams)", 7900)
            k("void jh::gemv_i8q4_kernel<2, 1, 1, 2, 0>(jh::GemvParams)", 4500)
            k("void jh::gemv_i8q4_kernel<1, 2, 2, 2, 1>(jh::GemvParams)", 15000)
            k("void jh::gemv_i8q4_kernel<2, 1, 1, 7, 0>(jh::GemvParams)", 9800)
        k("void jh::gemv_f32q4_kernel<4, 2, 2>(jh::GemvParams)", 65000)
        k("void jh::finish_token_kernel(float const*)", 3000)
    con.commit()
    con.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rocpd_timeline.py"), str(db), "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "6 replayed decode tokens in the trace, longest consecutive run 6" in out
    per_token = 2 * (5.8 + 7.9 + 4.5 + 15.0 + 9.8) + 65.0 + 3.0
    assert f"per token: {per_token:.1f} us inside kernels + {12 * 0.5:.1f} us between them" in out
    assert "| gemv_i8q4_kernel<1, 2, 2, 2, 1> | 2 | 15.00 | 0.50 |" in out


def test_reference_order_streaming_loops_do_not_copy_their_prefetch_ring():
    """tools/isa_ring_copies.py on the two units that hold the reference-order streaming GEMVs: no loop may end with a block of
    register moves behind s_waitcnt vmcnt(...) -- the shape hipcc gives a prefetch ring whose refills land in fresh registers, which
    makes every ring block wait out a full memory round trip (what bounded the few-row GEMVs until round 5)."""
    import subprocess
    import sys
    import shutil
    import pytest
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "isa_ring_copies.py")
    for unit in ("gemv_ref", "gemv_bf16"):
        r = subprocess.run([sys.executable, tool, unit], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        assert f"{unit}: no loop ends by copying a prefetch ring" in r.stdout, r.stdout[:2000]
    # the same ISA (isa_ring_copies.py leaves it in /tmp/_isa_<unit>.s): no register with a pending asm-issued LDS write is touched,
    # and none is pending at a label or a branch (tools/isa_pending_lds.py; the few-row GEMVs hold their activation operands that way)
    pend = os.path.join(root, "tools", "isa_pending_lds.py")
    r = subprocess.run([sys.executable, pend, "--file", "/tmp/_isa_gemv_ref.s"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "no register touched while its LDS write is pending" in r.stdout, r.stdout[-3000:]
    assert " 0 asm LDS reads" not in r.stdout, "the checker found no hand-pipelined read at all: " + r.stdout
    # and no streaming kernel drains its prefetch ring in front of the prologue barrier (tools/isa_ring_drain.py: a wait for fewer
    # loads than the ring holds between the fill and the barrier -- round 6: one register collision cost the o-projection 12 %)
    drain = os.path.join(root, "tools", "isa_ring_drain.py")
    for unit in ("gemv_ref", "gemv_bf16"):
        r = subprocess.run([sys.executable, drain, "--file", f"/tmp/_isa_{unit}.s"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and ", 0 drain their ring" in r.stdout, r.stdout[-3000:]
        assert " 0 streaming kernels checked" not in r.stdout


def test_ring_drain_checker_on_hand_made_isa(tmp_path):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "isa_ring_drain.py")
    ring = "\tglobal_load_dwordx4 v[4:7], v1, s[0:1] nt\n\tglobal_load_dword v8, v2, s[2:3] nt\n"

    def run(wait):
        f = tmp_path / "k.s"
        f.write_text("_ZN2jh20gemv_i8q4_p16_kernelILi2ELi1ELi8ELi4ELi512EEEvNS_10GemvParamsEii:\n\tglobal_load_dwordx4 v[20:23], v[10:11], off\n" + ring +
                     f"\ts_waitcnt vmcnt({wait})\n\tv_add_f32_e32 v3, v20, v3\n\ts_barrier\n\ts_waitcnt vmcnt(0)\n\ts_endpgm\n")
        return subprocess.run([sys.executable, tool, "--file", str(f)], capture_output=True, text=True)

    ok = run(2)            # waits for the activation load only: the two ring loads stay in flight
    assert ok.returncode == 0 and "1 streaming kernels checked, 0 drain" in ok.stdout, ok.stdout
    bad = run(0)
    assert bad.returncode == 1 and "vmcnt(0)" in bad.stdout, bad.stdout


def test_pending_lds_checker_flags_a_copy_in_front_of_the_wait(tmp_path):
    """The checker itself, on hand-made ISA: (1) the pattern hipcc produced at a control-flow merge -- a copy of the pending
    registers in front of the wait; (2) a read still pending at a branch; (3) the legal form."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pend = os.path.join(root, "tools", "isa_pending_lds.py")

    def run(body):
        f = tmp_path / "k.s"
        f.write_text("_Z1kv:\n" + body + "\ts_endpgm\n")
        return subprocess.run([sys.executable, pend, "--file", str(f)], capture_output=True, text=True)

    req = "\t;;#ASMSTART\n\tds_read_b128 v[4:7], v1\n\t;;#ASMEND\n"
    tie = "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n"
    bad_copy = run(req + "\tv_mov_b64_e32 v[8:9], v[4:5]\n" + tie + "\tv_add_f32_e32 v2, v8, v2\n")
    assert bad_copy.returncode == 1 and "v_mov_b64_e32" in bad_copy.stdout, bad_copy.stdout
    bad_branch = run(req + "\ts_cbranch_scc1 .LBB0_2\n" + tie + ".LBB0_2:\n")
    assert bad_branch.returncode == 1 and "s_cbranch_scc1" in bad_branch.stdout, bad_branch.stdout
    good = run(req + "\tv_add_f32_e32 v2, v3, v2\n" + tie + "\tv_add_f32_e32 v2, v4, v2\n\ts_cbranch_scc1 .LBB0_2\n.LBB0_2:\n")
    assert good.returncode == 0 and "1 asm LDS reads" in good.stdout, good.stdout
