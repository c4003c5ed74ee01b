"""Algorithmic bytes per ray from the kernel's own counters.

SURVEY's figure is  box_bytes * n_boxes + 48 * n_tris + 76 * p_shaded + 20 / r  with "wide nodes charged at their
real size / arity-equivalent".  n_boxes = child boxes tested per ray (trg_stats.node_fetches counts units of two
boxes).  The node layouts this build actually loads:

  * scene staged in LDS: sign-ordered BVH2 node, one step reads three 16-byte slab pairs + the 8-byte child pair
    = 56 bytes for two boxes                                                       -> 28 bytes per box
  * scene in HBM: quantised 4-wide node, 64 bytes for four boxes (q4node.h)          -> 16 bytes per box

(the first layouts of this project were 64 bytes per two boxes and 128 bytes per four = SURVEY's 32 bytes per box;
compressing the node lowers the bytes the algorithm has to move, so it lowers this figure too -- by design.)
48 = one triangle record, 76 = normals + colours + material id of a shaded hit, 20 = offset read + float4 write per
pixel sample."""

BOX_BYTES_LDS = 28.0
BOX_BYTES_HBM = 16.0


def algorithmic_bytes_per_ray(st, pixel_samples):
    rays = st.primary_rays + st.bounce_rays + st.shadow_rays
    n_boxes = 2.0 * st.node_fetches / rays
    n_tris = st.tri_tests / rays
    p_shaded = st.shaded_hits / rays
    rbar = rays / pixel_samples
    box = BOX_BYTES_LDS if st.scene_in_lds else BOX_BYTES_HBM
    total = box * n_boxes + 48.0 * n_tris + 76.0 * p_shaded + 20.0 / rbar
    return total, dict(nodes_per_ray=n_boxes, tris_per_ray=n_tris, shaded_per_ray=p_shaded, rays_per_pixel_sample=rbar,
                       bytes_per_box=box)
