// q8node.h -- the 80-byte COMPRESSED 8-WIDE BVH node (round-5 experiment, -DTRG_WIDE8=1; after Ylitie, Karras, Laine 2017, "Efficient
// Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs").  Up to eight child boxes quantised to 8 bits per plane against the
// node's own origin and per-axis power-of-two scale; the children sit in SLOTS whose index bits say on which side of the node's centre a
// child lies (bit 0: +x, bit 1: +y, bit 2: +z), so that a ray whose direction signs are `oct` meets the slots roughly in the order of
// slot ^ ~oct descending -- no sorting network.  The inner children of a node have consecutive node indices (child = child_base + the number
// of inner slots below this one), the leaves of a node own consecutive pairs of leaf records (leaf = rec_base + 2 x the number of leaf slots
// below this one): a traversal keeps GROUPS -- (child_base, hit mask) -- on its stack instead of single children.
//
//   dword 0..2   origin x, y, z (float)          dword 3   scale x (float, a power of two)
//   dword 4      child_base                      dword 5   rec_base (first leaf record of this node)
//   dword 6      imask | lmask << 8              (bit s: slot s is an inner node / a leaf; neither: empty, its box is inverted)
//   dword 7      the upper halves of scale y (low 16 bits) and scale z (high 16 bits): a power of two has nothing below them
//   dword 8..19  qlo.x[0..3] qlo.x[4..7] qlo.y[0..3] qlo.y[4..7] | qlo.z[..] qlo.z[..] qhi.x[..] qhi.x[..] | qhi.y[..] qhi.y[..] qhi.z[..] qhi.z[..]
//                one byte per slot; plane = origin + q * scale; lo rounded DOWN, hi UP (in double): the decoded box contains the float box
#pragma once
#include <stdint.h>

namespace trg {
constexpr uint32_t kQ8NodeBytes = 80, kQ8NodeDwords = 20;
constexpr uint32_t kRec8HasNext = 1u, kRec8Quad = 2u;   // Bvh::rec8_flags
}  // namespace trg
