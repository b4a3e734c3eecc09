/* pgd_oracle.h — entry points of the CPU oracle (TEST INFRASTRUCTURE; see pgd_oracle.c header). */
#ifndef PGD_ORACLE_H
#define PGD_ORACLE_H
#include <stdint.h>

#include "../include/pgdrive_hip.h"
#include "../include/pgd_state_layout.h"

typedef struct orc_engine* orc_handle;

#define ORC_NF PGD_NF
#define ORC_NI PGD_NI
#define ORC_NEI PGD_NEI

int orc_obs_dim(const pgd_config* c);
orc_handle orc_create(const pgd_config* cfg);
int orc_upload_maps(orc_handle h, const pgd_map* maps, int n_maps, const pgd_lane* lanes, int n_lanes,
                    const pgd_road* roads, int n_roads, const pgd_box* boxes, int n_boxes, const int32_t* cs, int n_cs,
                    const int32_t* ci, int n_ci);
int orc_upload_scenarios(orc_handle h, const pgd_scenario* scen, int n_scen, const pgd_spawn* spawns);
/* float64 values of the float fields of the tables / spawn records uploaded before (see pgd_oracle.c) */
int orc_upload_tables_f64(orc_handle h, const double* lanes, int n_lanes, const double* boxes, int n_boxes, const double* maps, int n_maps);
int orc_upload_spawns_f64(orc_handle h, const double* spawns, int n);
int orc_upload_config_f64(orc_handle h, const double* values, int n);
void orc_destroy(orc_handle h);
void orc_state_dims(int* nf, int* ni, int* nei);
void orc_get_state(orc_handle h, double* f, int32_t* i, int32_t* ei);
void orc_set_state(orc_handle h, const double* f, const int32_t* i, const int32_t* ei);
int orc_reset(orc_handle h, const int32_t* env_ids, const int32_t* scen_ids, int n, double* obs);
int orc_observe(orc_handle h, double* obs);
int orc_refresh(orc_handle h);
int orc_step(orc_handle h, const float* actions, double* obs, double* reward, uint8_t* done, uint32_t* flags);
int orc_step_mt(orc_handle h, int threads, const float* actions, double* obs, double* reward, uint8_t* done, uint32_t* flags);
int orc_run_mt(orc_handle h, int threads, int steps, const float* actions, int n_ring, double* obs, double* reward,
               uint8_t* done, uint32_t* flags);
int orc_step_range(orc_handle h, int e0, int e1, const float* actions, double* obs, double* reward, uint8_t* done,
                   uint32_t* flags);
uint32_t orc_rng(uint32_t seed, uint32_t a, uint32_t b, uint32_t c);
int orc_topdown_enable(orc_handle h, const pgd_topdown_config* c);
int orc_observe_topdown(orc_handle h, double* img /*[N,R,R,C]*/);
#endif
