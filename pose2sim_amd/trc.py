""".trc writer / reader, byte-compatible with the reference.

make_trc restates triangulation.py:151-215 (header lines :195-199, Z-up -> Y-up column
permutation of common.py:596-612, rows through ``DataFrame.to_csv(sep='\\t', header=None,
lineterminator='\\n')`` so that float formatting -- shortest repr, NaN as empty field -- is the
reference's own).  read_trc restates common.py:149-175.
"""
import glob
import logging
import os

import numpy as np
import pandas as pd


def zup2yup(Q):
    """common.py:596-612: (X, Y, Z) -> (Y, Z, X) per marker, on a DataFrame with 3N columns."""
    cols = list(Q.columns)
    cols = np.array([[cols[i * 3 + 1], cols[i * 3 + 2], cols[i * 3]] for i in range(int(len(cols) / 3))]).flatten()
    return Q[cols]


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


def make_trc(config_dict, Q, keypoints_names, id_person=-1):
    """triangulation.py:151-215.  Q: DataFrame, 3 columns (X, Y, Z in the Z-up frame) per keypoint,
    index = absolute frame numbers.  Returns the path written."""
    project_dir = config_dict.get('project').get('project_dir')
    multi_person = config_dict.get('project').get('multi_person')
    base = os.path.basename(os.path.realpath(project_dir))
    seq_name = f'{base}_P{id_person}' if multi_person else f'{base}'
    pose3d_dir = os.path.join(project_dir, 'pose-3d')
    frame_rate = resolve_frame_rate(config_dict)

    trc_f = f'{seq_name}_{Q.index[0]}-{Q.index[-1]}.trc'
    DataRate = CameraRate = OrigDataRate = frame_rate
    NumFrames = len(Q)
    NumMarkers = len(keypoints_names)
    header_trc = ['PathFileType\t4\t(X/Y/Z)\t' + trc_f,
                  'DataRate\tCameraRate\tNumFrames\tNumMarkers\tUnits\tOrigDataRate\tOrigDataStartFrame\tOrigNumFrames',
                  '\t'.join(map(str, [DataRate, CameraRate, NumFrames, NumMarkers, 'm', OrigDataRate, Q.index[0], NumFrames])),
                  'Frame#\tTime\t' + '\t\t\t'.join(keypoints_names) + '\t\t\t',
                  '\t\t' + '\t'.join([f'X{i + 1}\tY{i + 1}\tZ{i + 1}' for i in range(len(keypoints_names))]) + '\t']
    Q = zup2yup(Q)
    Q.insert(0, 't', Q.index / frame_rate)
    if not os.path.exists(pose3d_dir):
        os.mkdir(pose3d_dir)
    trc_path = os.path.realpath(os.path.join(pose3d_dir, trc_f))
    with open(trc_path, 'w') as trc_o:
        for line in header_trc:
            trc_o.write(line + '\n')
    # the data rows: DataFrame.to_csv(sep='\t', index=True, header=None, lineterminator='\n') of the reference
    # (:214), written by the native formatter (csrc/p2s_trc.cpp: repr() floats, NaN -> empty field)
    write_rows(trc_path, np.asarray(Q.index), Q.iloc[:, 0].to_numpy(), Q.iloc[:, 1:].to_numpy())
    return trc_path


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


def read_trc(trc_path):
    """common.py:149-175 -> (Q_coords, frames_col, time_col, markers, header)."""
    try:
        with open(trc_path, 'r') as trc_file:
            header = [next(trc_file) for _ in range(5)]
        markers = header[3].split('\t')[2::3]
        markers = [m.strip() for m in markers if m.strip()]
        trc_df = pd.read_csv(trc_path, sep='\t', skiprows=4, encoding='utf-8')
        frames_col, time_col = trc_df.iloc[:, 0], trc_df.iloc[:, 1]
        Q_coords = trc_df.drop(trc_df.columns[[0, 1]], axis=1)
        Q_coords = Q_coords.loc[:, ~Q_coords.columns.str.startswith('Unnamed')]
        Q_coords.columns = np.array([[m, m, m] for m in markers]).ravel().tolist()
        return Q_coords, frames_col, time_col, markers, header
    except Exception as e:
        raise ValueError(f'Error reading TRC file at {trc_path}: {e}')
