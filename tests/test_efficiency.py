"""f4: the efficiency table (reference lmms_eval/models/llava_onevision.py:65-77, 595-634)."""
import pytest
import torch

from vidcom2_amd.efficiency import EfficiencyMeter, format_efficiency_table

# what the reference prints for these three rows (layout captured from its formatter's behaviour: title, rule,
# header, rule, rows, rule; left-aligned cells padded to the widest entry of each column)
WANT = "\n".join([
    "Efficiency Analysis",
    "+--------------+---------+",
    "| Metric       | Value   |",
    "+--------------+---------+",
    "| LLM_time_s   | 12.346  |",
    "| Total_time_s | 100.000 |",
    "| Peak_mem_MB  | 20345.7 |",
    "+--------------+---------+",
])


def test_table_layout():
    rows = [("LLM_time_s", f"{12.3456:.3f}"), ("Total_time_s", f"{100:.3f}"), ("Peak_mem_MB", f"{20345.67:.1f}")]
    assert format_efficiency_table(rows) == WANT
    # short values: the header sets the column width
    t = format_efficiency_table([("a", "1")], title="T")
    assert t == "T\n+--------+-------+\n| Metric | Value |\n+--------+-------+\n| a      | 1     |\n+--------+-------+"


def test_meter_rows_without_a_device():
    m = EfficiencyMeter()
    m.total_cuda_time, m.max_mem = 12.3456, 20345.67
    assert m.table(wall_time=100.0) == WANT
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            with m.generation():
                pass


@pytest.mark.gpu
def test_meter_times_device_work_and_tracks_peak_memory():
    import time
    m = EfficiencyMeter()
    x = torch.randn(4096, 4096, device="cuda")
    for _ in range(2):
        with m.generation():
            y = x @ x
            big = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")     # 64 MiB inside the timed region
            del big
            with m.stage("Compress"):
                z = y.sum()
    torch.cuda.synchronize()
    assert m.calls == 2 and m.total_cuda_time > 0 and m.max_mem >= 64 + 2 * 64      # x, y (64 MiB each) + big
    rows = dict(m.rows())
    assert list(rows)[:3] == ["LLM_time_s", "Total_time_s", "Peak_mem_MB"] and "Compress_time_s" in rows
    assert float(rows["Compress_time_s"]) <= float(rows["LLM_time_s"]) + 1e-3
    time.sleep(0.01)
    assert float(dict(m.rows())["Total_time_s"]) >= float(rows["LLM_time_s"])
