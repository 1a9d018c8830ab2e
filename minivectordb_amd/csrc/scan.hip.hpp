// scan.hip.hpp -- int8 distance kernels of assembly/math.ts (dist_l2_i8 :115, dot_i8 :133)
// as a batched gather kernel, plus the exact brute-force scan (scan_mfma.hip.hpp).
#pragma once

#include <hip/hip_runtime.h>

#include <climits>

#include "device_index.hpp"
#include "traverse.hip.hpp"

namespace mvdb {

// One 8-lane group per (query, id) pair: 8 lanes x 16 B = 128 contiguous bytes per step.
// out = exact i32 L2 (metric 0) or dot (metric 1); ids < 0 -> INT32_MIN.
__global__ __launch_bounds__(256) void dist_pairs_kernel(DevIndexView ix, const int8_t *q, const int32_t *ids,
                                                         int per_q, long long pairs, int metric, int32_t *out)
{
	const int lane = threadIdx.x & 63;
	const int s = lane & 7;
	const long long pair = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
	const bool live = pair < pairs;
	const long long pc = live ? pair : 0;
	const int id = ids[pc];
	const int8_t *qrow = q + (size_t)(pc / per_q) * (size_t)ix.dim;
	const int8_t *vrow = ix.vec + (size_t)(id >= 0 ? id : 0) * (size_t)ix.row_stride;
	int qq = 0, vv = 0, qv = 0;
	for (int b0 = s * 16; b0 < ix.row_stride; b0 += 128) {
		const int4 v = *reinterpret_cast<const int4 *>(vrow + b0);
		const int vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int t = 0; t < 4; t++) {
			const int b = b0 + 4 * t;
			const int qw = b < ix.dim ? *reinterpret_cast<const int *>(qrow + b) : 0;
			qq = __builtin_amdgcn_sdot4(qw, qw, qq, false);
			vv = __builtin_amdgcn_sdot4(vw[t], vw[t], vv, false);
			qv = __builtin_amdgcn_sdot4(qw, vw[t], qv, false);
		}
	}
	const int part = metric == 0 ? qq + vv - 2 * qv : qv;
	const int tot = reduce8(part);
	if (live && s == 0) out[pair] = id >= 0 ? tot : INT_MIN;
}

}  // namespace mvdb

#include "scan_mfma.hip.hpp"
