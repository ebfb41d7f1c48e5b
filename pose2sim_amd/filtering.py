"""Filtering stage, Butterworth branch: mirror of Pose2Sim.filtering.filter_all for `[filtering] type = 'butterworth'`
(filtering.py:437-471, 728-830) with the per-column filtfilt loops replaced by one call into the HIP engine
(p2s_butterworth_host: every column of the .trc at once, one lane per column).

The coefficients come from the same SciPy calls the reference makes (scipy.signal.butter with order / 2 and
cut_off_frequency / (frame_rate / 2); scipy.signal.lfilter_zi for filtfilt's initial state), so the kernel reproduces
scipy.signal.filtfilt sample for sample.  The other filter types of the reference (kalman, gcv_spline, loess, ...)
depend on packages outside the hot path's scope and are refused with NotImplementedError.
"""
import glob
import logging
import os

import numpy as np

from . import trc as trc_mod

SUPPORTED_TYPES = ('butterworth',)


def butterworth_coefficients(order, cutoff, frame_rate):
    """(b, a, zi) exactly as filtering.py:453-457 builds them and scipy.signal.filtfilt initialises its passes."""
    from scipy import signal
    b, a = signal.butter(int(order) / 2, int(cutoff) / (frame_rate / 2), 'low', analog=False)
    return np.asarray(b, dtype=np.float64), np.asarray(a, dtype=np.float64), np.asarray(signal.lfilter_zi(b, a), dtype=np.float64)


def _make_engine():
    from .engine import Engine
    return Engine(int(os.environ.get('LOCAL_RANK', '0')))


def butterworth_filter(data, order, cutoff, frame_rate, engine=None):
    """Every column of data [n_frames][n_cols] through butterworth_filter_1d (filtering.py:437-471) on the GPU."""
    b, a, zi = butterworth_coefficients(order, cutoff, frame_rate)
    engine = engine or _make_engine()
    return engine.butterworth(np.asarray(data, dtype=np.float64), b, a, zi)


def _frame_rate(config_dict, project_dir):
    """filtering.py:762-774: the configured rate, or for 'auto' the first video's (60-fps warning text, 30 fps value,
    as in the reference)."""
    frame_rate = config_dict.get('project').get('frame_rate')
    if frame_rate != 'auto':
        return frame_rate
    video_files = glob.glob(os.path.join(project_dir, 'videos', '*' + config_dict['pose']['vid_img_extension']))
    try:
        rate = trc_mod.mp4_frame_rate(video_files[0])
        if not rate:
            raise ValueError
        return round(rate)
    except Exception:
        logging.warning('Cannot read video. Frame rate will be set to 60 fps.')
        return 30


def recap_filter3d(config_dict, trc_path):
    """The Butterworth lines of filtering.py:667-725."""
    fcfg = config_dict.get('filtering')
    bw = fcfg.get('butterworth')
    lines = ['--> Outliers rejected with a Hampel filter.' if fcfg.get('reject_outliers', False)
             else '--> No outlier rejection applied. Set reject_outliers to true in Config.toml to reject outliers.']
    if fcfg.get('filter', True):
        lines.append(f"--> Filter type: Butterworth low-pass. Order {int(bw.get('order'))}, Cut-off frequency {int(bw.get('cut_off_frequency'))} Hz.")
    else:
        lines.append('--> No filtering applied. Set filtering to true in Config.toml to filter coordinates.')
    lines.append(f'Filtered 3D coordinates are stored at {trc_path}.')
    for line in lines:
        logging.info(line)


def filter_all(config_dict, engine=None):
    """Same contract as the reference's filter_all for the Butterworth type: every `pose-3d/*.trc` without 'filt' in its
    path -> `<name>_<f0>-<f1>_filt_butterworth.trc` with the coordinates filtered column by column."""
    project_dir = config_dict.get('project').get('project_dir')
    pose3d_dir = os.path.realpath(os.path.join(project_dir, 'pose-3d'))
    fcfg = config_dict.get('filtering')
    do_filter = fcfg.get('filter', True)
    filter_type = fcfg.get('type')
    frame_range = config_dict.get('project').get('frame_range')
    if fcfg.get('reject_outliers', False):
        raise NotImplementedError('reject_outliers (Hampel filter) is outside the accelerated path')
    if do_filter and filter_type not in SUPPORTED_TYPES:
        raise NotImplementedError(f"filter type '{filter_type}' is outside the accelerated path; supported: {SUPPORTED_TYPES}")
    frame_rate = _frame_rate(config_dict, project_dir)
    bw = fcfg.get('butterworth')

    out_paths = []
    trc_path_in = [file for file in glob.glob(os.path.join(pose3d_dir, '*.trc')) if 'filt' not in file]
    for person_id, t_path_in in enumerate(trc_path_in):
        logging.info(f'\nFiltering 3D coordinates for person {person_id}...')
        t_file_in = os.path.basename(t_path_in)
        Q_coords, frames_col, time_col, markers, header = trc_mod.read_trc(t_path_in)

        first, last = int(frames_col.iloc[0]), int(frames_col.iloc[-1])          # filtering.py:786-792
        whole = frame_range in ('all', 'auto', []) or first > frame_range[0] or int(frames_col.iloc[1]) < frame_range[1]
        f_range = [first, last + 1] if whole else frame_range
        frame_nb = f_range[1] - f_range[0]
        lo = frames_col[frames_col == f_range[0]].index[0]
        hi = frames_col[frames_col == f_range[1] - 1].index[0] + 1
        Q_coords = Q_coords.iloc[lo:hi].reset_index(drop=True)
        frames_col = frames_col.iloc[lo:hi].reset_index(drop=True)
        time_col = time_col.iloc[lo:hi].reset_index(drop=True)

        t_path_out = t_path_in.replace(t_path_in.split('_')[-1], f'{f_range[0]}-{f_range[1]}_filt_{filter_type}.trc')
        t_file_out = os.path.basename(t_path_out)
        header[0] = header[0].replace(t_file_in, t_file_out)                     # :796-798
        header[2] = '\t'.join(part if i != 2 else str(frame_nb) for i, part in enumerate(header[2].split('\t')))
        header[2] = '\t'.join(part if i != 7 else str(frame_nb) + '\n' for i, part in enumerate(header[2].split('\t')))

        if not do_filter:
            logging.warning(f'reject_outliers and filter have been set to false. No further processing done on {t_path_in}.\n')
            continue
        data = Q_coords.to_numpy(dtype=np.float64)
        filtered = butterworth_filter(data, bw.get('order'), bw.get('cut_off_frequency'), frame_rate, engine)
        with open(t_path_out, 'w') as trc_o:
            trc_o.writelines(header)
        trc_mod.write_rows(t_path_out, frames_col.to_numpy(), time_col.to_numpy(), filtered)
        recap_filter3d(config_dict, t_path_out)
        out_paths.append(t_path_out)
    return out_paths
