"""Keypoint order, names and the left/right swap map of a pose model.

Mirrors what the reference derives from its skeleton trees at triangulation.py:716-749 and
personAssociation.py:695-711: pre-order traversal, nodes with id != None, then the name-prefix
rule R<->L / right<->left for the mirrored keypoint index.
"""
from .skeleton_tables import TABLES

# triangulation.py:717-723
ALIASES = {
    'BODY_WITH_FEET': 'HALPE_26',
    'WHOLE_BODY_WRIST': 'COCO_133_WRIST',
    'WHOLE_BODY': 'COCO_133',
    'BODY': 'COCO_17',
    'HAND': 'HAND_21',
    'FACE': 'FACE_106',
    'ANIMAL': 'ANIMAL2D_17',
}


def resolve_model_name(pose_model):
    up = str(pose_model).upper()
    return ALIASES.get(up, pose_model)


def _custom_rows(tree):
    """Flatten a [pose.CUSTOM]-style nested dict {name, id, children:[...]} in pre-order
    (anytree DictImporter semantics, triangulation.py:727-730)."""
    rows = []

    def walk(node, parent, is_root):
        me = len(rows)
        nid = node.get('id')
        if is_root and nid == 'None':
            nid = None
        rows.append((node['name'], nid, parent))
        for c in node.get('children', []) or []:
            walk(c, me, False)
    walk(tree, -1, True)
    return tuple(rows)


def model_rows(pose_model, config_dict=None):
    """Rows (name, id, parent_row) for a model name, alias or a custom tree in config_dict['pose']."""
    name = resolve_model_name(pose_model)
    if name in TABLES:
        return TABLES[name]
    try:
        return _custom_rows(config_dict.get('pose').get(pose_model))
    except Exception:
        # same message (unformatted braces included) as triangulation.py:732
        raise NameError('{pose_model} not found in skeletons.py nor in Config.toml')


def swap_indices(names):
    """triangulation.py:742-749."""
    try:
        sw = ['L' + n[1:] if n.startswith('R') else 'R' + n[1:] if n.startswith('L') else n for n in names]
        sw = [n.replace('right', 'left') if n.startswith('right') else
              n.replace('left', 'right') if n.startswith('left') else n for n in sw]
        return [names.index(n) for n in sw], True
    except Exception:
        return list(range(len(names))), False


def keypoints(pose_model, config_dict=None):
    """-> (keypoints_ids, keypoints_names, keypoints_idx_swapped)."""
    rows = model_rows(pose_model, config_dict)
    ids = [r[1] for r in rows if r[1] is not None]
    names = [r[0] for r in rows if r[1] is not None]
    swap, _ = swap_indices(names)
    return ids, names, swap


def node_id_by_name(pose_model, node_name, config_dict=None):
    """personAssociation.py:748: id of the first node called node_name (may be None)."""
    for name, nid, _ in model_rows(pose_model, config_dict):
        if name == node_name:
            return nid
    raise IndexError(node_name)
