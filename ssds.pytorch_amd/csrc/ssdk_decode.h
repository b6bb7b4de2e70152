// ssdk_decode.h -- host-side glue between the translation units of the decode stage:
//   ssdk_decode.hip  scan_kernel, level_kernel        ssdk_tail.hip  levelsel_kernel (per-level select + decode), nmswalk_kernel (NMS)
//   ssdk_scan16.hip  scan16_kernel (16-bit heads)
//   ssdk_nms.hip     nms_kernel                       ssdk_ctx.cpp   ssdk_decode_nms[_ctx], contexts, profiling
#pragma once
#include "ssdk_common.h"

namespace ssdk {

struct DecodePlan {  // how the scan is cut into units
  u32 tiles_per_unit, units_per_image;  // tiles_per_unit: the target size; tpu[l] what level l actually uses
  u32 units[SSDK_MAX_LEVELS], unit_base[SSDK_MAX_LEVELS], n[SSDK_MAX_LEVELS], tpu[SSDK_MAX_LEVELS];
  size_t cand_bytes, cnt_bytes;
  bool fused;  // the geometry fits levelsel_kernel + nmswalk_kernel (only asked for when ndet > 0); else level_kernel + nms_kernel
};

int make_plan(const ssdk_level* lv, int L, int B, int dtype, int K, DecodePlan* pl, int ndet);
int launch_scan(const ssdk_level* lv, int L, int B, int dtype, float thr, int K, const DecodePlan& pl, void* ws,
                size_t ws_bytes, hipStream_t stream, unsigned long long* stamps);
int launch_level(const ssdk_level* lv, int L, int B, int dtype, int K, int rescore, const DecodePlan& pl, const void* ws,
                 float* scores, float* boxes, float* classes, hipStream_t stream);
// ssdk_scan16.hip: the 16-bit / positive-threshold scan (what the plan is cut for whenever dtype and K allow it)
struct ScanParams;
bool scan16_applies(int dtype, float thr, int K);
u32 scan16_threshold_pattern(int dtype, float thr);
u32 scan16_max_tiles_per_unit(int K);
int launch_scan16(const ScanParams& sp, int dtype, int B, u32 units_per_image, hipStream_t stream);
void hist_window(float thr, u32* base, u32* shift);
size_t tail_fits(int K, int L, int ndet);
int launch_levelsel(const ssdk_level* lv, int L, int B, int dtype, int K, int rescore, const u32* units, const u32* unit_base,
                    u32 units_per_image, const void* cand, const void* cand_cnt, u32 hist_base, u32 hist_sh, float* ms,
                    float* mb, float* mc, unsigned long long* stamps, hipStream_t stream);
int launch_nmswalk(const float* ms, const float* mb, const float* mc, int B, int N, float nms_thr, int ndet, int diou, float* os,
                   float* ob, float* oc, unsigned long long* stamps, hipStream_t stream);
int launch_nms(const float* scores, const float* boxes, const float* classes, int B, int N, float thr, int ndet,
               int diou, float* os, float* ob, float* oc, hipStream_t stream);

}  // namespace ssdk
