// iss_mi355x.hip -- C-ABI shared library of the MI355X read-generation engine (see include/iss_mi355x.h).
// Host side: context, HBM uploads (model tables, genomes; small records from an arena), launch sequencing on one HIP
// stream for one record (iss_generate) or a whole work list (iss_generate_batch: records side by side in one arena),
// HIP-event timing, downloads, the FASTQ pipeline (text or gzip members built on the device, copy stream, writer thread).
// Device side: iss_kernels.hip.h (the Philox path), iss_mt_compat.hip.h (the reference's Mersenne-Twister streams),
// iss_fastq.hip.h, iss_deflate.hip.h.
#include "iss_mi355x.h"

#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "iss_kernels.hip.h"
#include "iss_fastq.hip.h"
#include "iss_deflate.hip.h"
#include "iss_mt_compat.hip.h"
#include "iss_units.hip.h"

// The host side by concern (one translation unit, one shared library; the order is the order of definition):
#include "iss_host_state.hip.h"       // FASTQ pipeline records, struct iss_ctx
#include "iss_host_util.hip.h"        // errors, uploads, switches, frees, kernel choice, timing, synchronisation
#include "iss_host_mt_streams.hip.h"  // MT19937 seeding and fill launches
#include "iss_host_fastq_pipe.hip.h"  // writer thread, flush
#include "iss_api_context.hip.h"
#include "iss_api_model.hip.h"
#include "iss_api_generate.hip.h"
#include "iss_api_mt.hip.h"
#include "iss_api_fastq.hip.h"
