"""tools/probes/pk_f32_victim.hip (packed vs scalar fp32, same operands) beside the synthetic co-runners of tools/probes/corun_stress.hip: which instruction of a
neighbouring wave makes v_pk_*_f32 return wrong values?  python tools/probes/pk_f32_victim2.py [trials]"""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
vic = ctypes.CDLL(os.path.join(here, "pk_f32_victim.so")); co = ctypes.CDLL(os.path.join(here, "corun_stress.so"))
vic.pk_victim_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
co.corun_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sink = torch.zeros(4, device="cuda"); buf = torch.zeros(64 << 20, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
names = {0: "fp16 MFMA 16x16x32 (VGPR acc)", 1: "fp32 MFMA 16x16x4", 2: "LDS hammer", 3: "packed fp32 VALU", 4: "global memory stream", 5: "v_cvt_pk_f16_f32", 6: "v_accvgpr_write/read",
         7: "fp16 MFMA 16x16x32 (AGPR acc)", 8: "fp16 MFMA + cvt_pk + pk_add", 9: "bf16 MFMA 16x16x32", 10: "fp16 MFMA 32x32x16", 11: "fp16 MFMA 16x16x16", 12: "fp16 MFMA x32 + v_pk_add_f32", 13: "fp16 MFMA x32 + v_cvt_pk_f16_f32", 14: "fp32 MFMA + v_pk_add_f32", 15: "fp16 MFMA 16x16x16 + v_pk_add_f32", 16: "fp16 MFMA x32 + v_add_f32", 17: "fp16 MFMA x32 + v_pk_mul_f32", 18: "fp16 MFMA x32 + cvt,cvt,pack", 19: "fp16 MFMA x32 + v_cvt_pkrtz_f16_f32", 20: "fp16 MFMA x32 + cvt, cvt_sdwa", 21: "fp16 MFMA 16x16x16 + v_cvt_pk_f16_f32", 22: "fp32 MFMA + v_cvt_pk_f16_f32", 23: "bf16 MFMA x32 + v_cvt_pk_bf16_f32", 24: "v_cvt_pk_f16_f32 + v_pk_add_f32", 25: "x32 MFMA waves beside cvt_pk waves", 26: "fp16 MFMA x32 + v_cvt_f32_f16", 27: "fp16 MFMA x32 + v_cvt_f32_u32", 28: "fp16 MFMA x32 + v_exp_f32", 29: "fp16 MFMA x32 + v_mul/v_fma_f32", 30: "fp16 MFMA x32, drained, + cvt_pk", -1: "nothing"}
for mode in [int(v) for v in os.environ.get('CORUN_MODES', '0,7,8,12,13,14,15,16,17,9,10,11,1,5,6,3,2,4,-1').split(',')]:
    bad = torch.zeros(1, dtype=torch.int32, device="cuda"); ex = torch.zeros(8, device="cuda")
    torch.cuda.synchronize()
    for t in range(trials):
        if mode >= 0: co.corun_launch(mode, 1024, 3000, sink.data_ptr(), buf.data_ptr(), buf.numel(), sa.cuda_stream)
        vic.pk_victim_launch(int(os.environ.get('VICTIM_FORM', '0')), 1024, 1500, 1024, bad.data_ptr(), ex.data_ptr(), sb.cuda_stream)
        torch.cuda.synchronize()
    print(f"co-runner {names[mode]:32s}: packed != scalar in {int(bad):9d} lane-rounds", [f"{v:.9g}" for v in ex.tolist()[:4]] if int(bad) else "")
