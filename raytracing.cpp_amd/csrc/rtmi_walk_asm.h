// rtmi_walk_asm.h -- the node steps of the BVH walk as hand-scheduled gfx950 loops, one per memory layout (included by
// rtmi_trace_kernel.h; the C++ node step beside them in the kernel serves the statistics and stamp variants).
#pragma once

#include "rtmi_kernel_common.h"

// The node steps of the walk as hand-scheduled loops (gfx950 ISA), one per memory layout.  A loop runs node steps for as long as
// the vote says "node" and more than `floor` lanes still walk, and returns the two counts of the trip it stopped at (the
// caller breaks or runs the leaf step).  Same arithmetic, instruction for instruction, as the C++ node step in the kernel
// (which stays the path of the STATS and RTMI_PROF variants) -- what is gone is the glue the structurizer puts around it:
// ~72 instructions per trip instead of ~95, and every instruction of this loop costs (10 extra per trip: 3-4 % of the frame).
//   48-byte node record: ctr[2][3] fp32 | half[2][3] fp16, rounded up on the host | child[2]
//   v66-v69 = c0x c0y c0z c1x   v70-v73 = c1y c1z h0x|h0y h0z|h1x   v74-v77 = h1y|h1z ch0 ch1 -
// v_fma_mix_f32 takes the fp16 half extents as they are (exact conversion inside the FMA: the same value as v_cvt_f32_f16 +
// v_fma_f32, six instructions less per trip).  References are signed (nodes >= 0, leaves < -1, sentinel -1); one stack level =
// `stride` bytes; the far child is stored above the top unconditionally and only counts when both boxes are hit.
// Inline asm is not seen by the hazard recogniser: no VALU-written mask is read by a VALU here (every mask a v_cndmask reads
// comes out of a scalar instruction), which is the one gfx950 hazard these sequences could meet.
//
// walk_nodes_hbm: trees that stay in HBM (config 4: 100k spheres), three 16-byte loads through L1 / L2 / Infinity Cache on an SGPR
// base, 32-bit stack entries.
// The first `ktop` nodes of the breadth-first numbering -- the levels every walk passes through -- are staged into LDS by every
// workgroup (the same 48-byte records at LDS address nbase + 48 * cur): a trip reads them with three ds_read_b128 and only
// the lanes below that top go to memory.  A node read through the vector-memory path costs the CU's address unit 16 cycles per
// 16-byte instruction whatever the hit rate (rocprofv3, round 3: 73 % of its cycles on config 4; round 4 with 384 nodes staged:
// vector-memory reads -45 % per frame together with the tighter pad, frame -3.8 % from the staging alone).
DEV void walk_nodes_hbm(Trav& t, const uint4* nodes, uint32_t nbase, uint32_t ktop, uint32_t stride, int floor, int& n_leaf, int& n_node) {
    int tmp;
    uint64_t m_node, m_leaf, saved, hit0, hit1;
    float x, y, z, tn0;
    asm volatile(
        "L_top_%=:\n\t"
        "v_cmp_le_i32_e64 %[mnode], 0, %[cur]\n\t"
        "v_cmp_gt_i32_e64 %[mleaf], -1, %[cur]\n\t"
        "s_bcnt1_i32_b64 %[nnode], %[mnode]\n\t"
        "s_bcnt1_i32_b64 %[nleaf], %[mleaf]\n\t"
        "s_add_i32 %[tmp], %[nnode], %[nleaf]\n\t"
        "s_cmp_le_i32 %[tmp], %[floor]\n\t"
        "s_cbranch_scc1 L_exit_%=\n\t"
        "s_cmp_gt_i32 %[nleaf], %[nnode]\n\t"
        "s_cbranch_scc1 L_exit_%=\n\t"
        "s_and_saveexec_b64 %[saved], %[mnode]\n\t"
        "v_mul_u32_u24_e32 %[x], 48, %[cur]\n\t"
        "v_cmp_gt_u32_e32 vcc, %[ktop], %[cur]\n\t"     // this lane's node is in the staged top
        "s_and_saveexec_b64 %[hit0], vcc\n\t"
        "s_cbranch_execz L_nolds_%=\n\t"
        "v_add_u32_e32 %[y], %[nbase], %[x]\n\t"
        "ds_read_b128 v[66:69], %[y]\n\t"
        "ds_read_b128 v[70:73], %[y] offset:16\n\t"
        "ds_read_b128 v[74:77], %[y] offset:32\n\t"
        "L_nolds_%=:\n\t"
        "s_andn2_b64 exec, %[hit0], vcc\n\t"
        "s_cbranch_execz L_nomem_%=\n\t"
        "global_load_dwordx4 v[66:69], %[x], %[nodes]\n\t"
        "global_load_dwordx4 v[70:73], %[x], %[nodes] offset:16\n\t"
        "global_load_dwordx4 v[74:77], %[x], %[nodes] offset:32\n\t"
        "L_nomem_%=:\n\t"
        "s_mov_b64 exec, %[hit0]\n\t"
        "s_waitcnt vmcnt(2) lgkmcnt(2)\n\t"
        "v_fma_f32 v66, v66, %[ix], %[ox]\n\t"      // tc0x
        "v_fma_f32 v67, v67, %[iy], %[oy]\n\t"      // tc0y
        "v_fma_f32 v68, v68, %[iz], %[oz]\n\t"      // tc0z
        "v_fma_f32 v69, v69, %[ix], %[ox]\n\t"      // tc1x
        "s_waitcnt vmcnt(1) lgkmcnt(1)\n\t"
        "v_fma_f32 v70, v70, %[iy], %[oy]\n\t"      // tc1y
        "v_fma_f32 v71, v71, %[iz], %[oz]\n\t"      // tc1z
        "v_fma_mix_f32 v78, v72, |%[ix]|, %[px] op_sel_hi:[1,0,0]\n\t"                  // th0x: the pad rides in the FMA of the half extent
        "v_fma_mix_f32 v72, v72, |%[iy]|, %[py] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"   // th0y
        "v_fma_mix_f32 v79, v73, |%[iz]|, %[pz] op_sel_hi:[1,0,0]\n\t"                  // th0z
        "v_fma_mix_f32 v73, v73, |%[ix]|, %[px] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"   // th1x
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "v_fma_mix_f32 v77, v74, |%[iy]|, %[py] op_sel_hi:[1,0,0]\n\t"                  // th1y
        "v_fma_mix_f32 v74, v74, |%[iz]|, %[pz] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"   // th1z
        "v_sub_f32_e32 %[x], v66, v78\n\t"          // box 0, near: max(x, y, max(z, 1e-4))
        "v_sub_f32_e32 %[y], v67, v72\n\t"
        "v_sub_f32_e32 %[z], v68, v79\n\t"
        "v_max_f32_e32 %[z], 0x38d1b717, %[z]\n\t"
        "v_max3_f32 %[tn0], %[x], %[y], %[z]\n\t"
        "v_add_f32_e32 %[x], v66, v78\n\t"          // far: min(x, y, min(z, tbest))
        "v_add_f32_e32 %[y], v67, v72\n\t"
        "v_add_f32_e32 %[z], v68, v79\n\t"
        "v_min_f32_e32 %[z], %[z], %[tbest]\n\t"
        "v_min3_f32 %[x], %[x], %[y], %[z]\n\t"
        "v_cmp_le_f32_e64 %[hit0], %[tn0], %[x]\n\t"
        "v_sub_f32_e32 %[x], v69, v73\n\t"          // box 1
        "v_sub_f32_e32 %[y], v70, v77\n\t"
        "v_sub_f32_e32 %[z], v71, v74\n\t"
        "v_max_f32_e32 %[z], 0x38d1b717, %[z]\n\t"
        "v_max3_f32 v66, %[x], %[y], %[z]\n\t"      // tn1
        "v_add_f32_e32 %[x], v69, v73\n\t"
        "v_add_f32_e32 %[y], v70, v77\n\t"
        "v_add_f32_e32 %[z], v71, v74\n\t"
        "v_min_f32_e32 %[z], %[z], %[tbest]\n\t"
        "v_min3_f32 %[x], %[x], %[y], %[z]\n\t"
        "v_cmp_le_f32_e64 %[hit1], v66, %[x]\n\t"
        "v_cmp_lt_f32_e32 vcc, v66, %[tn0]\n\t"     // nearer1
        "s_orn2_b64 vcc, vcc, %[hit0]\n\t"
        "s_and_b64 %[mleaf], %[hit1], vcc\n\t"      // take1 = hit1 & (!hit0 | nearer1): the nearer child first
        "v_cndmask_b32_e64 %[x], v76, v75, %[mleaf]\n\t"   // the far child: take1 ? ch0 : ch1
        "ds_write_b32 %[sp], %[x]\n\t"
        "v_cndmask_b32_e64 %[cur], v75, v76, %[mleaf]\n\t" // take1 ? ch1 : ch0
        "s_and_b64 vcc, %[hit0], %[hit1]\n\t"
        "v_cndmask_b32_e32 %[y], 0, %[stride], vcc\n\t"
        "v_add_u32_e32 %[sp], %[sp], %[y]\n\t"
        "s_or_b64 vcc, %[hit0], %[hit1]\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"           // neither box hit: pop
        "s_cbranch_execz L_nopop_%=\n\t"
        "v_sub_u32_e32 %[sp], %[sp], %[stride]\n\t"
        "ds_read_b32 %[cur], %[sp]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "L_nopop_%=:\n\t"
        "s_mov_b64 exec, %[saved]\n\t"
        "s_branch L_top_%=\n\t"
        "L_exit_%=:"
        : [cur] "+v"(t.cur), [sp] "+v"(t.sp), [nleaf] "=&s"(n_leaf), [nnode] "=&s"(n_node), [tmp] "=&s"(tmp),
          [mnode] "=&s"(m_node), [mleaf] "=&s"(m_leaf), [saved] "=&s"(saved), [hit0] "=&s"(hit0), [hit1] "=&s"(hit1),
          [x] "=&v"(x), [y] "=&v"(y), [z] "=&v"(z), [tn0] "=&v"(tn0)
        : [ix] "v"(t.inv.x), [iy] "v"(t.inv.y), [iz] "v"(t.inv.z), [ox] "v"(t.oinv.x), [oy] "v"(t.oinv.y),
          [oz] "v"(t.oinv.z), [px] "v"(t.pinv.x), [py] "v"(t.pinv.y), [pz] "v"(t.pinv.z), [tbest] "v"(t.tbest),
          [stride] "v"(stride), [nodes] "s"(nodes), [nbase] "s"(nbase), [ktop] "s"(ktop), [floor] "s"(floor)
        : "vcc", "scc", "memory", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77",
          "v78", "v79");
}

// walk_nodes_lds: trees staged into LDS whole (up to 8192 spheres), the same record at LDS address nbase + 48 * cur, 16-bit
// stack entries.  Rounds 1-3 kept 64-byte all-fp32 records here: four reads per trip and a stride of 16 dwords that put every
// read of a trip on 16 of the 64 banks.  A/B on MI355X (S-RTOW 1080p x 128 spp, profiles/r04_lds_node_layout.txt):
// SQ_LDS_BANK_CONFLICT 2.39 G -> 1.06 G (27 % -> 15 % of SQ_LDS_IDX_ACTIVE, itself -20 %), SQ_WAIT_INST_LDS -25 %, LDS
// 73.5 -> 68.9 KB per workgroup -- and the frame time unchanged within 0.2 %: the walk never waited on the banks.
DEV void walk_nodes_lds(Trav& t, uint32_t nbase, uint32_t stride, int floor, int& n_leaf, int& n_node) {
    int tmp;
    uint64_t m_node, m_leaf, saved, hit0, hit1;
    float x, y, z, tn0;
    asm volatile(
        "L_top_%=:\n\t"
        "v_cmp_le_i32_e64 %[mnode], 0, %[cur]\n\t"
        "v_cmp_gt_i32_e64 %[mleaf], -1, %[cur]\n\t"
        "s_bcnt1_i32_b64 %[nnode], %[mnode]\n\t"
        "s_bcnt1_i32_b64 %[nleaf], %[mleaf]\n\t"
        "s_add_i32 %[tmp], %[nnode], %[nleaf]\n\t"
        "s_cmp_le_i32 %[tmp], %[floor]\n\t"
        "s_cbranch_scc1 L_exit_%=\n\t"
        "s_cmp_gt_i32 %[nleaf], %[nnode]\n\t"
        "s_cbranch_scc1 L_exit_%=\n\t"
        "s_and_saveexec_b64 %[saved], %[mnode]\n\t"
        "v_mad_u32_u24 %[x], %[cur], 48, %[nbase]\n\t"
        "ds_read_b128 v[66:69], %[x]\n\t"
        "ds_read_b128 v[70:73], %[x] offset:16\n\t"
        "ds_read_b128 v[74:77], %[x] offset:32\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_fma_f32 v66, v66, %[ix], %[ox]\n\t"      // tc0x
        "v_fma_f32 v67, v67, %[iy], %[oy]\n\t"      // tc0y
        "v_fma_f32 v68, v68, %[iz], %[oz]\n\t"      // tc0z
        "v_fma_f32 v69, v69, %[ix], %[ox]\n\t"      // tc1x
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_fma_f32 v70, v70, %[iy], %[oy]\n\t"      // tc1y
        "v_fma_f32 v71, v71, %[iz], %[oz]\n\t"      // tc1z
        "v_fma_mix_f32 v78, v72, |%[ix]|, %[px] op_sel_hi:[1,0,0]\n\t"                  // th0x: the pad rides in the FMA of the half extent
        "v_fma_mix_f32 v72, v72, |%[iy]|, %[py] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"   // th0y
        "v_fma_mix_f32 v79, v73, |%[iz]|, %[pz] op_sel_hi:[1,0,0]\n\t"                  // th0z
        "v_fma_mix_f32 v73, v73, |%[ix]|, %[px] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"   // th1x
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_fma_mix_f32 v77, v74, |%[iy]|, %[py] op_sel_hi:[1,0,0]\n\t"                  // th1y
        "v_fma_mix_f32 v74, v74, |%[iz]|, %[pz] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"   // th1z
        "v_sub_f32_e32 %[x], v66, v78\n\t"          // box 0, near: max(x, y, max(z, 1e-4))
        "v_sub_f32_e32 %[y], v67, v72\n\t"
        "v_sub_f32_e32 %[z], v68, v79\n\t"
        "v_max_f32_e32 %[z], 0x38d1b717, %[z]\n\t"
        "v_max3_f32 %[tn0], %[x], %[y], %[z]\n\t"
        "v_add_f32_e32 %[x], v66, v78\n\t"          // far: min(x, y, min(z, tbest))
        "v_add_f32_e32 %[y], v67, v72\n\t"
        "v_add_f32_e32 %[z], v68, v79\n\t"
        "v_min_f32_e32 %[z], %[z], %[tbest]\n\t"
        "v_min3_f32 %[x], %[x], %[y], %[z]\n\t"
        "v_cmp_le_f32_e64 %[hit0], %[tn0], %[x]\n\t"
        "v_sub_f32_e32 %[x], v69, v73\n\t"          // box 1
        "v_sub_f32_e32 %[y], v70, v77\n\t"
        "v_sub_f32_e32 %[z], v71, v74\n\t"
        "v_max_f32_e32 %[z], 0x38d1b717, %[z]\n\t"
        "v_max3_f32 v66, %[x], %[y], %[z]\n\t"      // tn1
        "v_add_f32_e32 %[x], v69, v73\n\t"
        "v_add_f32_e32 %[y], v70, v77\n\t"
        "v_add_f32_e32 %[z], v71, v74\n\t"
        "v_min_f32_e32 %[z], %[z], %[tbest]\n\t"
        "v_min3_f32 %[x], %[x], %[y], %[z]\n\t"
        "v_cmp_le_f32_e64 %[hit1], v66, %[x]\n\t"
        "v_cmp_lt_f32_e32 vcc, v66, %[tn0]\n\t"     // nearer1
        "s_orn2_b64 vcc, vcc, %[hit0]\n\t"
        "s_and_b64 %[mleaf], %[hit1], vcc\n\t"      // take1 = hit1 & (!hit0 | nearer1): the nearer child first
        "v_cndmask_b32_e64 %[x], v76, v75, %[mleaf]\n\t"   // the far child: take1 ? ch0 : ch1
        "ds_write_b16 %[sp], %[x]\n\t"
        "v_cndmask_b32_e64 %[cur], v75, v76, %[mleaf]\n\t" // take1 ? ch1 : ch0
        "s_and_b64 vcc, %[hit0], %[hit1]\n\t"
        "v_cndmask_b32_e32 %[y], 0, %[stride], vcc\n\t"
        "v_add_u32_e32 %[sp], %[sp], %[y]\n\t"
        "s_or_b64 vcc, %[hit0], %[hit1]\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"           // neither box hit: pop
        "s_cbranch_execz L_nopop_%=\n\t"
        "v_sub_u32_e32 %[sp], %[sp], %[stride]\n\t"
        "ds_read_i16 %[cur], %[sp]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "L_nopop_%=:\n\t"
        "s_mov_b64 exec, %[saved]\n\t"
        "s_branch L_top_%=\n\t"
        "L_exit_%=:"
        : [cur] "+v"(t.cur), [sp] "+v"(t.sp), [nleaf] "=&s"(n_leaf), [nnode] "=&s"(n_node), [tmp] "=&s"(tmp),
          [mnode] "=&s"(m_node), [mleaf] "=&s"(m_leaf), [saved] "=&s"(saved), [hit0] "=&s"(hit0), [hit1] "=&s"(hit1),
          [x] "=&v"(x), [y] "=&v"(y), [z] "=&v"(z), [tn0] "=&v"(tn0)
        : [ix] "v"(t.inv.x), [iy] "v"(t.inv.y), [iz] "v"(t.inv.z), [ox] "v"(t.oinv.x), [oy] "v"(t.oinv.y),
          [oz] "v"(t.oinv.z), [px] "v"(t.pinv.x), [py] "v"(t.pinv.y), [pz] "v"(t.pinv.z), [tbest] "v"(t.tbest),
          [stride] "v"(stride), [nbase] "s"(nbase), [floor] "s"(floor)
        : "vcc", "scc", "memory", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77",
          "v78", "v79");
}
