// snac_tune.h -- the ids of the dispatch table's entries (snac_hip.hip KNOBS, same order) and tune(id): the effective value of an entry
// (its default or its environment override).  Internal; separate from snac_dev.h so that a new knob does not change the kernels' source.
#pragma once

namespace snac_detail {
enum Tn { TN_PIPELINE, TN_TILE, TN_2D_TILE32_MIN, TN_3D_BLOCK, TN_3D_BLOCK_MIN, TN_3D_BLOCK_MIN_F64, TN_3D_BLOCK_MIN_F32, TN_2D_STAGE, TN_2D_STAGE_MIN, TN_2D_STAGE_MIN_F64,
          TN_2D_STAGE_MIN_F32, TN_2D_TP, TN_2D_TP_MAX, TN_2D_TP_MAX_F64, TN_2D_TP_GAP_LO, TN_2D_TP_GAP_HI, TN_2D_TP_MAX_F32, TN_2D_TP_MAX_ODD, TN_2D_TP_VAR_MAX,
          TN_2D_TP_VAR_PLAN, TN_2D_TP_VAR_SHORT, TN_2D_TP_EB8, TN_1D_TP, TN_1D_TP_MAX, TN_1D_TP_MAX_F64, TN_1D_TP_MAX_F32, TN_1D_TP_VAR_MAX, TN_1D_TP_EB16,
          TN_STEP_STAGE, TN_STEP_VAR_MIN, TN_STEP_VAR_SHORT, TN_STEP_VAR_HALF_LO, TN_STEP_VAR_HALF_HI, TN_STEP_VAR_F64, TN_STEP_VAR_F32, TN_STEP_VAR_HALF,
          TN_STEP_VAR3_MIN, TN_STEP3D_SPAN, TN_STEP3D_SPAN_MIN, TN_T2D_E, TN_EDGES3D, TN_EDGES2D, TN_EDGES2D_MIN, TN_3D_BLOCK_VAR, TN_3D_BLOCK_VAR_MIN, TN_3D_BLOCK_VAR_PLAN_F64, TN_3D_BLOCK_VAR_PLAN_F32, TN_2D_BLOCK, TN_2D_BLOCK_MIN_F64, TN_2D_BLOCK_MAX_F64, TN_2D_BLOCK_MIN_F32,
          TN_2D_BLOCK_MAX_F32, TN_2D_BLOCK_TWO_F64, TN_2D_BLOCK_TWO_F32, TN_2D_BLOCK_VAR_MIN, TN_2D_BLOCK_VAR_MAX, TN_2D_BLOCK_VAR_TWO, TN_COUNT };
int tune(int id);

}  // namespace snac_detail
