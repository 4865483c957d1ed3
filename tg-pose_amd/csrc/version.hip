#include "tgp_common.h"
extern "C" int tgp_version(void) { return TGP_ABI_VERSION; }

// Node census of a captured hipGraph: counts[0..3] = kernel, memcpy, memset, every other node type.  Host-only (no launch).
// The trainer's step capture (autograd.GraphedStep) uses it as a guard: a MEMSET node in a captured step means an ATen
// multi-block reduction (its semaphores are zeroed with hipMemsetAsync) slipped into the capture -- see DESIGN.md section 3.
extern "C" int tgp_graph_node_counts(void *graph, int *counts)
{
    if (!graph || !counts) return TGP_EINVAL;
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    size_t n = 0;
    hipError_t e = hipGraphGetNodes((hipGraph_t)graph, nullptr, &n);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return 0;
    hipGraphNode_t *nodes = new hipGraphNode_t[n];
    e = hipGraphGetNodes((hipGraph_t)graph, nodes, &n);
    for (size_t i = 0; e == hipSuccess && i < n; ++i) {
        hipGraphNodeType t;
        e = hipGraphNodeGetType(nodes[i], &t);
        if (e != hipSuccess) break;
        if (t == hipGraphNodeTypeKernel) ++counts[0];
        else if (t == hipGraphNodeTypeMemcpy) ++counts[1];
        else if (t == hipGraphNodeTypeMemset) ++counts[2];
        else ++counts[3];
    }
    delete[] nodes;
    return (int)e;
}

#ifdef TGP_DEV
// development library only (scripts/capture_memset_probe.py): hipMemsetAsync on the caller's stream, so that a Python probe can
// put a MEMSET node between two kernel nodes of a captured graph and test how hipGraph orders it
extern "C" int tgp_debug_memset_async(void *ptr, int value, size_t bytes, tgp_stream_t stream)
{
    return (int)hipMemsetAsync(ptr, value, bytes, tgp_hs(stream));
}
#endif
