// TEST INFRASTRUCTURE (not part of the product): lets the host-logic unit tests run where no GPU exists by re-pointing the
// C-ABI names of include/bsgpu.h at the CPU oracle (oracle/libbs_oracle.so exports the same entry points with prefix bso_).
// Force-included ahead of everything by tests/test_host_cpp.py (-include tests/host/oracle_backend.h); the declarations of
// include/bsgpu.h then declare the bso_ symbols.
#pragma once
#define bsgpu_abi_version bso_abi_version
#define bsgpu_add_factors bso_add_factors
#define bsgpu_add_factors_indirect bso_add_factors_indirect
#define bsgpu_add_marginal bso_add_marginal
#define bsgpu_sync_factors_indirect bso_sync_factors_indirect
#define bsgpu_clear bso_clear
#define bsgpu_covariance bso_covariance
#define bsgpu_create bso_create
#define bsgpu_create_error bso_create_error
#define bsgpu_dense_solve bso_dense_solve
#define bsgpu_destroy bso_destroy
#define bsgpu_evaluate bso_evaluate
#define bsgpu_finalize bso_finalize
#define bsgpu_get_blocks bso_get_blocks
#define bsgpu_get_iteration bso_get_iteration
#define bsgpu_get_marginal bso_get_marginal
#define bsgpu_last_error bso_last_error
#define bsgpu_marginalize bso_marginalize
#define bsgpu_nconst bso_nconst
#define bsgpu_nidx bso_nidx
#define bsgpu_nres bso_nres
#define bsgpu_num_iterations_recorded bso_num_iterations_recorded
#define bsgpu_num_parameters_tangent bso_num_parameters_tangent
#define bsgpu_num_residuals bso_num_residuals
#define bsgpu_options_default bso_options_default
#define bsgpu_options_vio bso_options_vio
#define bsgpu_plan_info bso_plan_info
#define bsgpu_preintegrate bso_preintegrate
#define bsgpu_reproj_jacobian_bytes bso_reproj_jacobian_bytes
#define bsgpu_reprojection_errors bso_reprojection_errors
#define bsgpu_reset_values bso_reset_values
#define bsgpu_set_blocks bso_set_blocks
#define bsgpu_set_cameras bso_set_cameras
#define bsgpu_set_values bso_set_values
#define bsgpu_solve bso_solve
#define bsgpu_tangent_offset bso_tangent_offset
#define bsgpu_time_reproj_jacobian_ms bso_time_reproj_jacobian_ms
#define bsgpu_triangulate bso_triangulate
