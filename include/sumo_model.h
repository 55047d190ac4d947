/* sumo_model.h -- binary layout of a compiled RoboSumo scene ("model blob").
 *
 * Written by robosumo_selfplay_amd/mjcf.py::SumoModel.to_blob(); read by the HIP engine
 * (robosumo_selfplay_amd/csrc) and by the CPU oracle (oracle/).  It is the MI355X-side stand-in for the
 * constant part of MuJoCo's mjModel that the reference obtains from mj_loadXML
 * (reference: mujoco-py/mujoco_py/cymj.pyx:162-176; field inventory mujoco-py/mujoco_py/pxd/mjmodel.pxd:444-600).
 *
 * Layout:  int32 n_int, int32 n_flt, int32 ints[n_int], float64 flts[n_flt]
 *   ints[0..15]  header {magic, version, nq, nv, nu, nbody, njnt, ngeom, npair, nagent, frame_skip,
 *                timestep_limit, 0,0,0,0}; then the int tables in SUMO_INT_TABLES order;
 *   flts[0]      tatami_size; then the float tables in SUMO_FLT_TABLES order.
 * Table lengths are functions of the header dims (second macro argument).
 *
 * The struct keeps offsets (not pointers) so the same value is valid for a host copy and a device copy of
 * the blob: set ibase/fbase to wherever the two sections live.
 */
#ifndef SUMO_MODEL_H
#define SUMO_MODEL_H

#include <stdint.h>
#include <stddef.h>

#define SUMO_BLOB_MAGIC 0x4F4D5553
#define SUMO_BLOB_VERSION 3

/* mjtGeom / mjtJoint values used (mujoco-py/mujoco_py/pxd/mjmodel.pxd:134-143, 123-128) */
#define SUMO_GEOM_PLANE 0
#define SUMO_GEOM_SPHERE 2
#define SUMO_GEOM_CAPSULE 3
#define SUMO_GEOM_CYLINDER 5
#define SUMO_GEOM_BOX 6
#define SUMO_JNT_FREE 0
#define SUMO_JNT_HINGE 3

/* X(name, length-expression in terms of d = dims) */
#define SUMO_INT_TABLES(X) \
  X(body_parentid, d->nbody) X(body_rootid, d->nbody) X(body_weldid, d->nbody) X(body_jntnum, d->nbody) \
  X(body_jntadr, d->nbody) X(body_dofnum, d->nbody) X(body_dofadr, d->nbody) X(body_geomadr, d->nbody + 1) \
  X(jnt_type, d->njnt) X(jnt_qposadr, d->njnt) X(jnt_dofadr, d->njnt) X(jnt_bodyid, d->njnt) \
  X(jnt_limited, d->njnt) X(dof_bodyid, d->nv) X(dof_jntid, d->nv) X(dof_parentid, d->nv) \
  X(geom_type, d->ngeom) X(geom_bodyid, d->ngeom) X(geom_condim, d->ngeom) X(actuator_dofid, d->nu) \
  X(pair_geom1, d->npair) X(pair_geom2, d->npair) \
  X(agent_qposadr, d->nagent) X(agent_nq, d->nagent) X(agent_dofadr, d->nagent) X(agent_nv, d->nagent) \
  X(agent_bodyadr, d->nagent) X(agent_nbody, d->nagent) X(agent_uadr, d->nagent) X(agent_nu, d->nagent) \
  X(agent_torso, d->nagent)

#define SUMO_FLT_TABLES(X) \
  X(opt, 8) X(qpos0, d->nq) X(body_pos, 3 * d->nbody) X(body_quat, 4 * d->nbody) X(body_ipos, 3 * d->nbody) \
  X(body_iquat, 4 * d->nbody) X(body_mass, d->nbody) X(body_inertia, 3 * d->nbody) \
  X(body_subtreemass, d->nbody) X(body_invweight0, 2 * d->nbody) X(jnt_pos, 3 * d->njnt) \
  X(jnt_axis, 3 * d->njnt) X(jnt_range, 2 * d->njnt) X(jnt_margin, d->njnt) X(dof_armature, d->nv) \
  X(dof_damping, d->nv) X(dof_invweight0, d->nv) X(geom_size, 3 * d->ngeom) X(geom_pos, 3 * d->ngeom) \
  X(geom_quat, 4 * d->ngeom) X(geom_rbound, d->ngeom) X(geom_friction, 3 * d->ngeom) X(geom_margin, d->ngeom) \
  X(geom_gap, d->ngeom) X(geom_solref, 2 * d->ngeom) X(geom_solimp, 5 * d->ngeom) X(geom_solmix, d->ngeom) \
  X(actuator_gear, d->nu) X(actuator_ctrlrange, 2 * d->nu) X(pair_margin, d->npair) X(pair_gap, d->npair) \
  X(pair_friction, 3 * d->npair) X(pair_solref, 2 * d->npair) X(pair_solimp, 5 * d->npair)

/* opt[] slots */
#define SUMO_OPT_TIMESTEP 0
#define SUMO_OPT_GRAVITY 1 /* 1,2,3 */
#define SUMO_OPT_TOLERANCE 4
#define SUMO_OPT_MEANINERTIA 5
#define SUMO_OPT_IMPRATIO 6
#define SUMO_OPT_ITERATIONS 7

typedef struct sumo_model {
  int nq, nv, nu, nbody, njnt, ngeom, npair, nagent, frame_skip, timestep_limit;
  double tatami_size;
  const int32_t* ibase; /* start of ints[] */
  const double* fbase;  /* start of flts[] */
#define X(name, len) int o_##name;
  SUMO_INT_TABLES(X)
  SUMO_FLT_TABLES(X)
#undef X
} sumo_model_t;

#define SUMO_I(m, name) ((m)->ibase + (m)->o_##name)
#define SUMO_F(m, name) ((m)->fbase + (m)->o_##name)

/* Parse a blob held in host memory.  Returns 0 on success, negative on a malformed blob.  On success
 * m->ibase / m->fbase point into `blob`; the caller may re-point them at a device copy. */
static inline int sumo_model_parse(sumo_model_t* m, const void* blob, size_t nbytes) {
  const int32_t* w = (const int32_t*)blob;
  if (nbytes < 8 + 16 * 4) return -1;
  int n_int = w[0], n_flt = w[1];
  if (n_int < 16 || n_flt < 1 || (size_t)8 + (size_t)n_int * 4 + (size_t)n_flt * 8 != nbytes) return -2;
  const int32_t* ints = w + 2;
  if (ints[0] != SUMO_BLOB_MAGIC || ints[1] != SUMO_BLOB_VERSION) return -3;
  m->nq = ints[2]; m->nv = ints[3]; m->nu = ints[4]; m->nbody = ints[5]; m->njnt = ints[6];
  m->ngeom = ints[7]; m->npair = ints[8]; m->nagent = ints[9]; m->frame_skip = ints[10];
  m->timestep_limit = ints[11];
  m->ibase = ints;
  m->fbase = (const double*)(ints + n_int);
  m->tatami_size = m->fbase[0];
  const sumo_model_t* d = m;
  int io = 16, fo = 1;
#define X(name, len) m->o_##name = io; io += (int)(len);
  SUMO_INT_TABLES(X)
#undef X
#define X(name, len) m->o_##name = fo; fo += (int)(len);
  SUMO_FLT_TABLES(X)
#undef X
  if (io > n_int || fo != n_flt) return -4;
  return 0;
}

#endif /* SUMO_MODEL_H */
