// capi_fastx.hip -- FASTQ / FASTA -> device batches: spans entry points, device indexer, streaming driver
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"
#include "util_kernels.hpp"

using namespace ntamd;
using namespace ntamd::host;

#include "fastx_stream.hpp"
