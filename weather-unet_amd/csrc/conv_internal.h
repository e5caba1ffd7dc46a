// internal: production conv3x3 path (conv3x3_mfma_v2.hip), dispatched from wu_conv3x3_fwd
#pragma once
#include <hip/hip_runtime.h>

bool conv_v2_eligible(int H, int W, int ldx, int Cin, int Cout, int stride, int dtype, bool masked);
// pool != NULL (act must be ReLU, no gate): the epilogue also writes the 2x2 max-pool of y
int conv_v2_launch(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy,
                   const void* egate, int ldegate, int egate_act, int N, int H, int W, int Cin, int Cout, int act, hipStream_t s,
                   void* pool = nullptr, int ldpool = 0);
// the same conv as TWO 4-wave workgroups per CU with one 80-KiB LDS buffer each (conv3x3_mfma_v2s.hip); same eligibility
int conv_v2s_launch(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy,
                    const void* egate, int ldegate, int egate_act, int N, int H, int W, int Cin, int Cout, int act, hipStream_t s,
                    void* pool = nullptr, int ldpool = 0);
// which of the two: option WU_OPT_CONV_V2S (0 = never, 1 = always, 2 = layers with Cin <= 128, 3 = Cin <= 64)
bool conv_use_v2s(int Cin);
