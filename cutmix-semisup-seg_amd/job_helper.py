"""
Mirror of the reference's job_helper.py: the `@job(name)` decorator that gives a training function a
`.submit(**kwargs)` entry point, creates `results/<job>/log_<desc>.txt` + `results/<job>/<desc>/`, tees
stdout/stderr into the log and skips a job whose log already exists (job_helper.py:28-146). Host-side plumbing only.
"""
import os
import re
import sys

LOG_PREFIX = re.compile(r'log_(\d+)')
JOB_DIR_PREFIX = re.compile(r'(\d+)')


class LogAlreadyExistsError(Exception):
    pass


class JobNotRun(Exception):
    """Raised by a job function that refuses to start (e.g. no dataset pipeline and no --synthetic): the log this
    attempt created is removed again -- otherwise a corrected rerun would be skipped as 'already executed' -- and the
    process exits non-zero."""


class Logger(object):
    """File + stream tee (appends, like the reference)."""

    def __init__(self, path, stream):
        self.path = path
        self.stream = stream

    def write(self, x):
        with open(self.path, 'a+') as f_out:
            f_out.write(x)
        self.stream.write(x)

    def flush(self):
        self.stream.flush()


class SubmitConfig(object):
    def __init__(self, job_name, job_desc, enumerate_job_names):
        res_dir = os.path.join('results', job_name)
        os.makedirs(res_dir, exist_ok=True)
        if job_desc == 'none':
            self.log_path = None
            self.job_out_dir = None
        elif enumerate_job_names:
            index = 0
            for name in os.listdir(res_dir):
                for rx in (LOG_PREFIX, JOB_DIR_PREFIX):
                    m = rx.match(name)
                    if m is not None:
                        index = max(index, int(m.group(1)) + 1)
            self.log_path = os.path.join(res_dir, 'log_{:04d}_{}.txt'.format(index, job_desc))
            self.job_out_dir = os.path.join(res_dir, '{:04d}_{}'.format(index, job_desc))
        else:
            self.log_path = os.path.join(res_dir, 'log_{}.txt'.format(job_desc))
            self.job_out_dir = os.path.join(res_dir, job_desc)
            if os.path.exists(self.log_path) or os.path.exists(self.job_out_dir):
                raise LogAlreadyExistsError
        self._run_dir = None
        if self.log_path is not None:
            self._stdout = Logger(self.log_path, sys.stdout)
            self._stderr = Logger(self.log_path, sys.stderr)

    @property
    def run_dir(self):
        if self._run_dir is None and self.job_out_dir is not None:
            self._run_dir = self.job_out_dir
            os.makedirs(self._run_dir, exist_ok=True)
        return self._run_dir

    def connect_streams(self):
        if self.log_path is not None:
            sys.stdout = self._stdout
            sys.stderr = self._stderr

    def disconnect_streams(self):
        if self.log_path is not None:
            sys.stdout = self._stdout.stream
            sys.stderr = self._stderr.stream


def job(job_name, enumerate_job_names=True):
    """Decorator: `fn.submit(job_desc=..., **kwargs)` runs `fn(submit_config, **kwargs)` under the job's log."""

    def decorate(job_fn):
        def run_job(**kwargs):
            specific = kwargs.pop('job_name', None) or job_name
            quota_group = kwargs.pop('quota_group', None)
            if quota_group is not None and quota_group != '':
                raise ValueError('quota_group not supported when dnnlib is not available')
            desc = kwargs.pop('job_desc', None)
            if desc is None or desc == '':
                desc = specific
            # One process per GPU (torchrun): only rank 0 owns the log / run directory; the other ranks run with the
            # reference's job_desc='none' semantics (no log file, no "already executed" test) so that every rank
            # enters the collectives of the training function.
            if int(os.environ.get('RANK', '0')) != 0:
                desc = 'none'
            try:
                submit_config = SubmitConfig(specific, desc, enumerate_job_names)
            except LogAlreadyExistsError:
                print('Job {}:{} already executed; skipping'.format(specific, desc))
                if int(os.environ.get('WORLD_SIZE', '1')) > 1:
                    # the other ranks are already on their way into init_process_group: do not leave them hanging
                    raise SystemExit('Job {}:{} already executed (rank 0 of a multi-process launch)'.format(specific, desc))
                return
            print('[NO dnnlib] logging to {}'.format(submit_config.log_path))
            submit_config.connect_streams()
            try:
                job_fn(submit_config, **kwargs)
            except JobNotRun as e:
                submit_config.disconnect_streams()
                if submit_config.log_path is not None and os.path.exists(submit_config.log_path):
                    os.remove(submit_config.log_path)
                raise SystemExit(str(e))
            finally:
                submit_config.disconnect_streams()

        job_fn.submit = run_job
        return job_fn

    return decorate
