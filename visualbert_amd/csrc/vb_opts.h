// vb_opts.h -- per-stream launch options (include/visualbert_hip.h: vb_stream_set_opts).  The table lives in misc.hip;
// every extern "C" entry point that launches kernels reads its stream's entry once, at entry.
#pragma once
#include "../../include/visualbert_hip.h"

// options attached to `stream` (all-zero defaults when the stream has no entry)
vb_stream_opts vb_opts_for(void* stream);

// scratch attached to `stream` (vb_stream_set_scratch); {NULL, 0} when the stream has none: kernels that would need it are not chosen
struct vb_scratch { void* ptr; int64_t bytes; };
vb_scratch vb_scratch_for(void* stream);
// layout: [VB_SCRATCH_COUNTER_BYTES of int32 arrival counters, zero between launches][fp32 partial-result slabs]
#define VB_SCRATCH_COUNTER_BYTES 16384
