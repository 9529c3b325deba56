// Compile-time tunables of the ray kernels (rtx_kernels.hip), all in one place.  Each can be overridden on the hipcc command
// line (-DRTX_WAVES=6: tools/build_variants.sh builds such variants for A/B runs); the defaults are what the product ships.
// Everything that used to be an experiment SWITCH (alternative walks, filters, stores ...) has been retired: the losing
// branches are kept as a patch (tools/research/r04_experiment_branches.patch, DESIGN_HISTORY.md).
#pragma once

#ifndef RTX_DBG
#define RTX_DBG 0               // 1: wave-level stage counters + per-wave pass-1 timeline, 2: + sampled outcomes / histograms (slow), 3: state-machine timers
#endif
#ifndef RTX_WAVE_TRACE
#define RTX_WAVE_TRACE 0        // 1: three timestamps per pass-1 wave (first pop, end of the last tile, busy ticks) in a PRODUCT build: tools/wave_tail.py
#endif
#ifndef RTX_WAVES
#define RTX_WAVES 5             // waves per SIMD of the general pass-1 kernels (512 / RTX_WAVES VGPRs; their 31.7 KB of LDS hold five blocks per CU): 4 loses 11-20 %; the PLAIN kernels: RTX_WAVES_PLAIN
#endif
#ifndef RTX_WAVES_PLAIN
#define RTX_WAVES_PLAIN 6       // waves per SIMD of the PLAIN pass-1 kernels: their LDS fits six blocks per CU; 80 VGPRs and 48 B of scratch, 3-4 % faster than 5 (89 VGPRs, no scratch at all)
#endif
#ifndef RTX_WAVES_SSAA
#define RTX_WAVES_SSAA 4        // the SSAA launch lasts as long as its slowest wave: fewer, unspilled waves (128 VGPRs)
#endif
#ifndef RTX_WAVES_FRAME
#define RTX_WAVES_FRAME 4       // rtxFrameKernel runs where the frame is bounded by its slowest work items: likewise
#endif
#ifndef RTX_WAVES_ANALYTIC
#define RTX_WAVES_ANALYTIC 4    // scenes without meshes: the whole castRay state in registers (128 VGPRs)
#endif
#ifndef RTX_POP_MANY
#define RTX_POP_MANY 4u         // pass 1: tiles taken per atomic in the cheap half of a queue (profiles/r04_ab_pop.txt: 8 / 16 / guided helpings lost)
#endif
#ifndef RTX_PRIO_TICKS
#define RTX_PRIO_TICKS 50000u   // pass 1: tiles that took more than 0.5 ms (100 MHz ticks) in the previous launch run at raised wave priority
#endif
#ifndef RTX_MAX_SPLITS
#define RTX_MAX_SPLITS 4        // halvings of a wide bundle (traceWave)
#endif
#ifndef RTX_LEAF_BATCH
#define RTX_LEAF_BATCH 1        // leaves noted before their references are processed: ONE where a launch is bound by throughput (pass 1) ...
#endif
#ifndef RTX_LEAF_BATCH_FEW
#define RTX_LEAF_BATCH_FEW 2    // ... TWO where it lasts as long as its slowest wave's chain of fetches (SSAA items, frame kernel): profiles/r04_ab_leaf_batch.txt
#endif
#ifndef RTX_SSAA_VERY
#define RTX_SSAA_VERY 2u        // x the "heavy" threshold of pass-1 time (knob ssaa_heavy_ticks): tiles above get 4-pixel SSAA waves
#endif
#ifndef RTX_SSAA_SPREAD_PX
#define RTX_SSAA_SPREAD_PX 4u   // pixels per wave for the tiles that were very slow in pass 1 (rtxSsaaCountKernel)
#endif
#ifndef RTX_BUNDLE_PAIRS
#define RTX_BUNDLE_PAIRS 1      // makeBundle: 1 = waveMaxMin (two interleaved DPP chains per coordinate).  Rounds 2-4 shipped 0 without meaning to: the switch was
#endif                          // tested (line 477) before it was defined (line 671) -- found when the switches were retired in round 5; A/B in profiles/r05_ab_*.txt
#ifndef RTX_FAST_INVLEN
#define RTX_FAST_INVLEN 1       // Vec3::normalize's (float)(1 / sqrt((double)len2)) through the exact fp32 fast path (rtx_kernels.hip, invLenD); 0 = always the fp64 expression
#endif
#ifndef RTX_SAME_ORIGIN
#define RTX_SAME_ORIGIN 1       // makeBundle: bundles whose rays all start at the camera take their origin box as that point (no reductions over the origins)
#endif
#ifndef RTX_REC2_RELOAD
#define RTX_REC2_RELOAD 1       // traceWave: the mesh's bundle-split constants are fetched again per bundle instead of living in SGPRs across the walk
#endif
#ifndef RTX_PRUNE_AXIS
#define RTX_PRUNE_AXIS 1        // node visit: the prune / plane records of a wide node are evaluated with lanes = (record, axis) -- pruneEval8 -- instead of lanes = records (16 of 64 lanes)
#endif
#ifndef RTX_WIDE_NOCULL
#define RTX_WIDE_NOCULL 1       // options::useBackfaceCulling = 0 takes the wide walk with prune records too (0: the stackless binary walk, as until round 5)
#endif
#ifndef RTX_FILTER_MASKS
#define RTX_FILTER_MASKS 1      // bundle filter: verdicts as wave masks (a ballot per compare, combined on the scalar side) instead of bools (whose ballots compiled to v_cndmask + v_cmp)
#endif
#ifndef RTX_EXACT_HOIST
#define RTX_EXACT_HOIST 1       // exact tests of a pass: "did a lane's t improve" is one compare after the survivors, not one per survivor (0: the per-survivor form, with its copies of t)
#endif
#ifndef RTX_ADVANCE_UNIFORM
#define RTX_ADVANCE_UNIFORM 1   // castRay state machine: advance() steps all lanes in one loop with a uniform exit (finished lanes sit out) instead of per-lane returns
#endif
#ifndef RTX_ONE_ADVANCE
#define RTX_ONE_ADVANCE 1       // castRayWave: advance() once per round, at its head (0: before the loop and at the end of every round -- two copies of its code)
#endif
#ifndef RTX_PRUNE_LANEK
#define RTX_PRUNE_LANEK 1       // pruneEval8: a lane's two addresses (PruneBlock words, LDS axis record) packed in one register per walk instead of 12 VALU instructions per visit
#endif
#ifndef RTX_TILE_RECOMPUTE
#define RTX_TILE_RECOMPUTE 1    // rtxPass1Kernel: a tile's pixel coordinates are derived from the list entry again after castRayWave (fresh parameters) instead of being kept across it
#endif
