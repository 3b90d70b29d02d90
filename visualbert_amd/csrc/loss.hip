#include "vb_rt.h"
#include "../../include/visualbert_hip.h"
