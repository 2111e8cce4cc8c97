"""Setup-side finite-element builders (meshes, RT0/P0 operators, level hierarchies).

Host-only numpy/scipy code that produces the CSR operators the C-ABI consumes; it stands in
for the MFEM/ParELAG setup calls of the reference drivers and is outside every timed region."""
from .mesh import (Mesh, box_mesh, build_faces, element_volumes, kuhn_cube_tet, mesh_from_json,  # noqa: F401
                   read_mfem_mesh,
                   refine_uniform)
from .problems import (DarcyLevel, DarcyProblem, Hierarchy, SamplerLevel, SamplerProblem,  # noqa: F401
                       build_darcy_problem, build_hierarchy, build_sampler_problem,
                       elements_near_points, l2_projection_ops, matern_coefficient)
from .hybrid import HybridLevel, HybridSamplerProblem, build_hybrid_sampler_problem  # noqa: F401
from .darcy_hybrid import DarcyHybridLevel, darcy_hybrid_level  # noqa: F401
from .rt0 import build_spaces, mass_contributions, mass_matrix, prolongation_p0  # noqa: F401
from .transfer import (box_intersection_gt, clipped_intersection_gt, intersection_gt,  # noqa: F401
                       l2_projection_hierarchy)
from .output import (compute_l2_error, compute_max_error, prolongate_to_fine_grid, read_gridfunction_p0,  # noqa: F401
                     save_field_glvis, save_mesh_glvis, write_mfem_mesh)
