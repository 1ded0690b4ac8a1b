"""The measurement tooling behind profiles/: plan-order attribution of PMC counters and of the rocprofv3 kernel trace
(tools/summarize_evidence.py, tools/align_trace.py) on synthetic CSVs -- every op of a launch plan is one dispatch, the
cls-row `_prefix_kernel` launches of MViT's pooling ops are not ops of their own, partial replays are ignored."""
import csv
import io
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

PER_OP = """  op stem.conv|8x4 c3->8                      0.1000 ms   100.0 GB/s   1.00 TF/s
  op conv_b.dw+se|8x4 c8 k3x3x3 psum          0.2000 ms   100.0 GB/s   1.00 TF/s
  op attn.pool_q|8x4 c8                        0.0500 ms   100.0 GB/s   1.00 TF/s
  op conv_c|8x4 c8->8 gate                     0.3000 ms   100.0 GB/s   1.00 TF/s
  conv_c           n=  1    0.300 ms     100.0 GB/s     1.00 TF/s
"""
KERNELS = ["stem7_kernel<1>", "dw3_plane_kernel<1, 2, 0>", "dw_prefix_kernelIDF16bEE", "dw3_plane_kernel<2, 2, 0>", "pw_stream_kernel<8>"]


def _write(path, header, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(header)
        w.writerows(rows)


def test_pmc_traffic_is_attributed_by_plan_order(tmp_path):
    import summarize_evidence as SE
    per_op = tmp_path / "per_op.txt"
    per_op.write_text(PER_OP)
    for name, ctr, base in (("fetch", "FETCH_SIZE", 10.0), ("write", "WRITE_SIZE", 1.0)):
        rows, disp = [], 0
        for rep in range(3):                           # three full replays ...
            for k, kern in enumerate(KERNELS):
                disp += 1
                rows.append([disp, kern, ctr, base * (k + 1)])
        disp += 1
        rows.append([disp, KERNELS[0], ctr, 999.0])    # ... and a truncated fourth one
        d = tmp_path / name
        d.mkdir()
        _write(d / (name + "_counter_collection.csv"), ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"], rows)
    got = SE.by_plan_order(str(tmp_path), str(per_op))
    # op 2 (attn.pool_q) is the 4th dispatch: the prefix kernel in front of it is skipped
    assert set(got) == {"stem.conv", "conv_b", "attn.pool_q", "conv_c"}
    assert got["stem.conv"]["hbm_bytes_per_launch"] == round((2 * 10.0 * 1 + 1.0 * 1) * 1024)
    assert got["conv_b"]["hbm_bytes_per_launch"] == round((2 * 10.0 * 2 + 1.0 * 2) * 1024)
    assert got["attn.pool_q"]["hbm_bytes_per_launch"] == round((2 * 10.0 * 4 + 1.0 * 4) * 1024)
    assert got["conv_c"]["hbm_bytes_per_launch"] == round((2 * 10.0 * 5 + 1.0 * 5) * 1024)
    assert got["conv_c"]["dispatches_profiled"] == 3
    # ... and per op: algorithmic bytes (GB/s x ms of the per-op table) beside the PMC bytes of the same dispatch, worst excess first
    rows = SE.per_op_traffic(str(tmp_path), str(per_op))
    assert len(rows) == 4
    alg = {"stem.conv": 10.0, "conv_b": 20.0, "attn.pool_q": 5.0, "conv_c": 30.0}       # MB: 100 GB/s x ms
    hbm = {"stem.conv": 21 * 1024 * 1, "conv_b": 21 * 1024 * 2, "attn.pool_q": 21 * 1024 * 4, "conv_c": 21 * 1024 * 5}   # bytes
    first = rows[0]
    assert first.startswith("| `attn.pool_q|")           # 0.086 MB against 5 MB: the smallest deficit sorts first
    for name in alg:
        row = [r for r in rows if r.startswith("| `" + name)][0]
        cells = [c.strip() for c in row.strip().strip("|").rsplit("|", 6)]       # (the op label itself contains a '|')
        assert abs(float(cells[2]) - alg[name]) < 0.06 and abs(float(cells[5]) - hbm[name] / 1e6) < 0.06
        assert abs(float(cells[6]) - hbm[name] / 1e6 / alg[name]) < 0.01


def test_kernel_trace_is_aligned_with_the_plan(tmp_path):
    import align_trace as AT
    per_op = tmp_path / "per_op.txt"
    per_op.write_text(PER_OP)
    rows, t = [], 1000
    for rep in range(2):
        for k, kern in enumerate(KERNELS):
            dur = 1000 * (k + 1)                        # ns
            rows.append([kern, t, t + dur])
            t += dur + 500
    trace = tmp_path / "trace_kernel_trace.csv"
    _write(trace, ["Kernel_Name", "Start_Timestamp", "End_Timestamp"], rows)
    buf = io.StringIO()
    with redirect_stdout(buf):
        AT.main(str(per_op), str(trace))
    out = buf.getvalue()
    assert "avg of 2 replays" in out
    line = [l for l in out.splitlines() if l.startswith("| conv_c |")][0]
    assert "| 1 | 0.0050 | 0.3000 |" in line and "pw_stream_kernel<8>" in line       # 5000 ns, not the prefix kernel's 3000
    line = [l for l in out.splitlines() if l.startswith("| attn.pool_q |")][0]
    assert "0.0040" in line and "dw3_plane_kernel<2, 2, 0>" in line


def test_design_md_measured_table_is_the_generators_output():
    """DESIGN.md section 4's table is GENERATED from profiles/r6/ (tools/design_table.py): every row of the generator's output
    stands verbatim in the document, and the traffic figures it quotes are the ones profiles/traffic.json holds for the commit
    stamped there -- the numbers a reader checks against profiles/ are the numbers in the text."""
    import json
    import design_table
    buf = io.StringIO()
    argv = sys.argv
    sys.argv = ["design_table.py", "r6"]
    try:
        with redirect_stdout(buf):
            design_table.main()
    finally:
        sys.argv = argv
    rows = [l for l in buf.getvalue().split("\n") if l.startswith("| ") and "clips/s (default form)" not in l]
    assert len(rows) == 4
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for r in rows:
        assert r in design, r[:80]
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert set(tj["_measured_on"]) >= {"x3d_m", "mvit_b_32x3", "slowfast_r50", "x3d_l"}
    for wl in ("x3d_m", "mvit_b_32x3", "slowfast_r50", "x3d_l"):
        line = json.load(open(os.path.join(ROOT, "profiles", "r6", wl + "_bench_default.json")))
        assert line["roofline"]["kernel"] in tj[wl]["_by_kernel"]     # the dominant symbol has its own PMC population
        assert line["cpu_baseline"]["kind"] in ("port", "reference") and len(line["cpu_baseline"]["threads_sweep"]) >= 2
