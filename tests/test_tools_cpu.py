"""Developer tools that post-process measurement files (no GPU): tools/shape_delta.py (per-shape delta of two shape tables)
and tools/trace_last_step.py (per-kernel time of the last marked step of a rocprofv3 kernel trace)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HEAD = "family                     M      N      K gg R V act  calls     ms   TFLOP/s    GB/s   bound-ms  of-bound\n"


def _row(fam, M, N, K, R, calls, ms):
    return f"{fam:20s} {M:7d} {N:6d} {K:6d}  0 {R} 0   0 {calls:6d} {ms:7.2f} {1000:9d} {1000:8d} {1.0:9.2f} {0.5:8.2f}\n"


def test_shape_delta_sorts_by_delta_and_sums_families(tmp_path):
    a, b = tmp_path / "a.txt", tmp_path / "b.txt"
    a.write_text(HEAD + _row("gemm_linear", 460800, 320, 320, 1, 7, 1.50) + _row("layernorm", 460800, 320, 0, 0, 28, 3.00) +
                 _row("gemm_conv3x3", 460800, 320, 2880, 1, 7, 5.80))
    b.write_text(HEAD + _row("gemm_linear", 460800, 320, 320, 1, 7, 2.40) + _row("layernorm", 460800, 320, 0, 0, 28, 4.00) +
                 _row("gemm_conv3x3", 460800, 320, 2880, 1, 7, 5.81))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shape_delta.py"), str(a), str(b), "unit test"],
                         capture_output=True, text=True, check=True).stdout
    assert "totals: A 10.30 ms, B 12.21 ms, delta +1.91 ms" in out
    fam = out.split("by family:")[1].split("by shape")[0]
    assert fam.index("layernorm") < fam.index("gemm_linear") < fam.index("gemm_conv3x3")          # +1.00, +0.90, +0.01
    shapes = out.split("by shape (sorted by delta):")[1]
    assert "gemm_conv3x3" not in shapes and shapes.index("layernorm") < shapes.index("gemm_linear")   # |delta| < 0.02 dropped


def test_trace_last_step_cuts_at_the_last_mark(tmp_path):
    rows = ["Kind,Agent_Id,Queue_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,End_Timestamp"]
    t = 1000

    def k(name, dur):
        nonlocal t
        rows.append(f'KERNEL_DISPATCH,1,1,1,"{name}",1,{t},{t + dur}')
        t += dur + 10
    k("void init_kernel()", 500)
    for step in range(3):
        k("void at::native::(anonymous namespace)::spin_kernel(long)", 50)
        k("void (anonymous namespace)::wgrad_pp_kernel<1>(WpArgs)", 300 + step)
        k("void (anonymous namespace)::wgrad_pp_kernel<1>(WpArgs)", 300 + step)
        k("void (anonymous namespace)::ln_bwd_kernel<1>(float)", 100)
    src, out_csv = tmp_path / "trace.csv", tmp_path / "out.csv"
    src.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_last_step.py"), str(src), str(out_csv)],
                         capture_output=True, text=True, check=True).stdout
    assert "last step: 3 kernels" in out
    body = out_csv.read_text().splitlines()
    assert body[0].startswith("Name,Calls,TotalDurationNs")
    assert "wgrad_pp_kernel<1>" in body[1] and ",2,604," in body[1]            # the last step's two launches: 302 + 302
    assert "ln_bwd_kernel" in body[2] and ",1,100," in body[2]
    assert not any("init_kernel" in ln or "spin_kernel" in ln for ln in body)
