#!/bin/bash
# Timing builds of the pair sampler (WRONG RESULTS, never shipped: alt_so/ is scratch) against the
# product build, interleaved on one box (scripts/gpu_ab.sh): what records in bin order could buy.
#   proto_accept          the accept decision from the uniform alone at the benchmark's rate -- the
#                         reference point for the two below (same events whatever records are picked)
#   proto_accept_noperm   + a list position taken as the record's own place: no list entry loaded
#   proto_accept_nolast   + the lists' ends not requested with the pair
# usage (on the GPU box): bash scripts/gpu_proto.sh <tag>
tag=${1:-proto}
# (one timed step behind one warm-up step of a fresh population: the wrong picks of a timing build make its
# populations drift away from the benchmark's within a few steps)
REPS=${REPS:-4} bash scripts/gpu_ab.sh $tag --steps 1 --warmup 1
cp gpurun_out/ab_$tag.txt gpurun_out/proto_$tag.txt
