"""bench.py's host-side helpers: which profiled kernel a roofline is for, and the arithmetic behind roofline.valu_issue."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_frame_kernel_picks_the_instance_that_rendered_the_timed_frames():
    names = ["void rtx::k_render_persistent<true, true, false, true, false>(rtx::DRenderParams const*)",
             "void rtx::k_render_persistent<false, true, false, true, false>(rtx::DRenderParams const*)",
             "void rtx::k_render_resident<false>(rtx::DRenderParams const*)", "rtx::k_scale(double const*, unsigned long, double, double*)"]
    assert bench.frame_kernel(names, bench.KERNEL_RESIDENT) == names[2]
    assert bench.frame_kernel(names, bench.KERNEL) == names[1]          # never the work-counting instance
    assert bench.frame_kernel(names[:1] + names[3:], bench.KERNEL) is None
    assert bench.short_kernel(names[2]) == "k_render_resident<false>"


def test_valu_issue_arithmetic():
    c = {"GRBM_GUI_ACTIVE": 8.0e6, "SQ_INSTS_VALU": 1.0e9, "SQ_INSTS_VALU_ADD_F64": 1.0e8, "SQ_INSTS_VALU_FMA_F64": 1.0e8, "SQ_INSTS_VALU_MUL_F64": 1.0e8,
         "SQ_INSTS_VALU_TRANS_F64": 1.0e7, "SQ_INSTS_VALU_INT64": 5.0e7, "SQ_INSTS_VALU_CVT": 5.0e7, "SQ_THREAD_CYCLES_VALU": 1.6e11,
         "SQ_ACTIVE_INST_VALU": 4.0e9, "SQ_WAVE_CYCLES": 1.0e10, "SQ_WAIT_ANY": 2.5e9, "SQ_WAIT_INST_ANY": 3.0e9, "SQ_INSTS_SALU": 4.0e8}
    v = bench.valu_issue(c, 2.0e7)
    simd_cycles = 1024 * 1.0e6
    slow, trans = 4.0e8, 1.0e7
    fast = 1.0e9 - slow - trans
    assert v["frac"] == round((4 * slow + 16 * trans + 2.2 * fast) / simd_cycles, 3)
    assert v["frac_if_every_instruction_took_4_cycles"] == round(4.0e9 / simd_cycles, 3)
    assert v["lanes_active"] == 0.625 and v["wave_cycles_waiting"] == 0.25 and v["wave_cycles_ready_but_not_issued"] == 0.3
    assert v["vector_instructions_per_ray"] == 50.0 and v["scalar_instructions_per_ray"] == 20.0
    assert bench.valu_issue({}, 1.0) is None                           # a pass that failed leaves the field empty, not wrong


def test_every_configuration_names_a_scene():
    from raytracing_amd import scenes as S

    for name, cfg in bench.CONFIGS.items():
        assert name in S.EXAMPLES or name.startswith("synthetic_1M_")
        assert cfg["build"] in ("host", "device") and cfg["run_spp"] <= cfg["spp"]
