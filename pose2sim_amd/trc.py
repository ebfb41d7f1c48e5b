""".trc writer / reader, byte-compatible with the reference.

write_trc / make_trc produce the file of triangulation.py:151-215: the five header lines (:195-199, here a table of
labels and facts), the Z-up -> Y-up column order of common.py:596-612, and the rows of ``DataFrame.to_csv(sep='\\t',
header=None, lineterminator='\\n')`` -- shortest-repr floats, NaN as an empty field -- from the native formatter.
load_trc / read_trc read such a file back (the reader contract of common.py:149-175).
"""
import glob
import logging
import os

import numpy as np
import pandas as pd


def yup_columns(n_markers):
    """Column order of a .trc row for data held (X, Y, Z) in the Z-up frame: the reference's zup2yup (common.py:596-612)
    writes (Y, Z, X) of every marker."""
    return (3 * np.arange(n_markers)[:, None] + np.array([1, 2, 0])[None, :]).ravel()


def mp4_frame_rate(path):
    """Frame rate of an MP4/MOV from its 'mvhd'/'mdhd' + 'stts' boxes, rounded like
    round(cap.get(CAP_PROP_FPS)) at triangulation.py:184.  Returns None when unreadable.
    (The reference asks OpenCV, which this image does not have.)"""
    try:
        with open(path, 'rb') as f:
            data = f.read(64 * 1024 * 1024)
        i = data.find(b'mdhd')
        while i >= 0:
            ver = data[i + 4]
            if ver == 1:
                timescale = int.from_bytes(data[i + 24:i + 28], 'big')
                duration = int.from_bytes(data[i + 28:i + 36], 'big')
            else:
                timescale = int.from_bytes(data[i + 16:i + 20], 'big')
                duration = int.from_bytes(data[i + 20:i + 24], 'big')
            j = data.find(b'stts', i)
            k = data.find(b'vmhd', i)
            nxt = data.find(b'mdhd', i + 4)
            is_video = k >= 0 and (nxt < 0 or k < nxt)
            if is_video and j >= 0 and timescale > 0 and duration > 0:
                n_entries = int.from_bytes(data[j + 8:j + 12], 'big')
                frames = 0
                for e in range(n_entries):
                    frames += int.from_bytes(data[j + 12 + 8 * e:j + 16 + 8 * e], 'big')
                if frames > 0:
                    return round(frames * timescale / duration)
            i = nxt
    except Exception:
        return None
    return None


def resolve_frame_rate(config_dict):
    """triangulation.py:173-187.  'auto' reads the first video; when that fails the reference warns
    about 60 fps and uses 30 -- reproduced."""
    project_dir = config_dict.get('project').get('project_dir')
    frame_rate = config_dict.get('project').get('frame_rate')
    if frame_rate == 'auto':
        video_dir = os.path.join(project_dir, 'videos')
        vid_img_extension = config_dict['pose']['vid_img_extension']
        video_files = glob.glob(os.path.join(video_dir, '*' + vid_img_extension))
        fps = mp4_frame_rate(video_files[0]) if video_files else None
        if not fps:
            logging.warning('Cannot read video. Frame rate will be set to 60 fps.')
            fps = 30
        frame_rate = fps
    return frame_rate


# The five header lines of a .trc file (triangulation.py:195-199), as a table: the labels of line 2 and how the values of
# line 3 are made from the run's facts.
_HEADER_FIELDS = (('DataRate', 'rate'), ('CameraRate', 'rate'), ('NumFrames', 'n_frames'), ('NumMarkers', 'n_markers'),
                  ('Units', 'units'), ('OrigDataRate', 'rate'), ('OrigDataStartFrame', 'first_frame'), ('OrigNumFrames', 'n_frames'))


def header_lines(file_name, marker_names, frame_rate, first_frame, n_frames):
    facts = {'rate': frame_rate, 'n_frames': n_frames, 'n_markers': len(marker_names), 'units': 'm', 'first_frame': first_frame}
    axes = [f'{axis}{i}' for i in range(1, len(marker_names) + 1) for axis in 'XYZ']
    return ['\t'.join(('PathFileType', '4', '(X/Y/Z)', file_name)),
            '\t'.join(label for label, _ in _HEADER_FIELDS),
            '\t'.join(str(facts[key]) for _, key in _HEADER_FIELDS),
            'Frame#\tTime\t' + ''.join(name + '\t\t\t' for name in marker_names),
            '\t\t' + ''.join(axis + '\t' for axis in axes)]


def write_trc(pose3d_dir, seq_name, frames, coords_zup, marker_names, frame_rate):
    """One .trc file `<seq_name>_<first>-<last>.trc` from frames [F] (absolute numbers) and coords_zup [F][3 K] (X, Y, Z
    per marker, Z up): the file make_trc (triangulation.py:151-215) writes.  Returns its real path."""
    frames = np.asarray(frames)
    coords_zup = np.asarray(coords_zup, dtype=np.float64)
    file_name = f'{seq_name}_{frames[0]}-{frames[-1]}.trc'
    if not os.path.exists(pose3d_dir):
        os.mkdir(pose3d_dir)
    trc_path = os.path.realpath(os.path.join(pose3d_dir, file_name))
    with open(trc_path, 'w') as fh:
        fh.write('\n'.join(header_lines(file_name, marker_names, frame_rate, frames[0], len(frames))) + '\n')
    # rows `frame <tab> frame / rate <tab> coordinates`, floats as repr() and NaN as an empty field (what DataFrame.to_csv
    # gives the reference), by the native formatter csrc/p2s_trc.cpp
    write_rows(trc_path, frames, frames / frame_rate, coords_zup[:, yup_columns(len(marker_names))])
    return trc_path


def make_trc(config_dict, Q, keypoints_names, id_person=-1):
    """The reference's entry (triangulation.py:151-215): Q is a DataFrame of 3 columns (X, Y, Z, Z up) per keypoint
    indexed by absolute frame number.  Returns the path written."""
    project_dir = config_dict.get('project').get('project_dir')
    seq_name = os.path.basename(os.path.realpath(project_dir))
    if config_dict.get('project').get('multi_person'):
        seq_name += f'_P{id_person}'
    return write_trc(os.path.join(project_dir, 'pose-3d'), seq_name, np.asarray(Q.index), Q.to_numpy(dtype=np.float64), keypoints_names,
                     resolve_frame_rate(config_dict))


def write_rows(trc_path, frames, time_col, data):
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    frames = np.ascontiguousarray(frames)
    if not np.issubdtype(frames.dtype, np.integer):
        raise ValueError('frame numbers must be integers')
    frames = frames.astype(np.int64, copy=False)
    time_col = np.ascontiguousarray(time_col, dtype=np.float64)
    data = np.ascontiguousarray(data, dtype=np.float64)
    n_rows = len(frames)
    if time_col.shape != (n_rows,) or data.ndim != 2 or data.shape[0] != n_rows:
        raise ValueError('frames, time and data disagree on the number of rows')
    ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None    # noqa: E731
    _lib.check(lib.p2s_trc_append_rows(os.fsencode(trc_path), n_rows, data.shape[1], ptr(frames), ptr(time_col), ptr(data), 0))


def load_trc(trc_path):
    """-> (frames int64 [F], time [F], coords [F][3 K] as stored (Y-up), marker names, the 5 header lines).  The number
    block goes through pandas' C tokenizer, the parser the reference reads it with (common.py:163), so that a value
    written back is the digit string it was read from."""
    with open(trc_path, 'r') as fh:
        header = [next(fh) for _ in range(5)]
    markers = [m.strip() for m in header[3].split('\t')[2::3] if m.strip()]
    block = pd.read_csv(trc_path, sep='\t', skiprows=5, header=None, encoding='utf-8')
    n_coord = 3 * len(markers)
    coords = block.iloc[:, 2:2 + n_coord].to_numpy(dtype=np.float64)
    if coords.shape[1] != n_coord:
        raise ValueError(f'{n_coord} coordinate columns expected for {len(markers)} markers, found {coords.shape[1]}')
    return block.iloc[:, 0].to_numpy(), block.iloc[:, 1].to_numpy(dtype=np.float64), coords, markers, header


def read_trc(trc_path):
    """The reference's reader contract (common.py:149-175) -> (Q_coords, frames_col, time_col, markers, header): a
    DataFrame with every marker's name on its three columns, two Series, the marker names, the header lines."""
    try:
        frames, time_col, coords, markers, header = load_trc(trc_path)
        Q_coords = pd.DataFrame(coords, columns=[m for m in markers for _ in range(3)])
        return Q_coords, pd.Series(frames), pd.Series(time_col), markers, header
    except Exception as e:
        raise ValueError(f'Error reading TRC file at {trc_path}: {e}')
