"""The reference's `lib.*` call surface (main/lib of SaeedRahmani/MPC_for_AV_at_Intersection), backed by the HIP
kernels of libmpcx.so.  Module and symbol names follow the reference so that its scenario scripts can shadow
`lib.mpc`, `lib.motion_primitive_search*`, `lib.collision_avoidance`, ... with these modules."""
