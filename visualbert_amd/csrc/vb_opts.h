// vb_opts.h -- per-stream launch options (include/visualbert_hip.h: vb_stream_set_opts).  The table lives in misc.hip;
// every extern "C" entry point that launches kernels reads its stream's entry once, at entry.
#pragma once
#include "../../include/visualbert_hip.h"

// options attached to `stream` (all-zero defaults when the stream has no entry)
vb_stream_opts vb_opts_for(void* stream);
