"""`dsc.profile()` — mirror of python/dsc/profiler.py:14-63 on top of dsc_traces_record / dsc_dump_traces /
dsc_clear_traces.  The dump holds two tracks: the operator calls on the host and the spans their kernels took on the
HIP stream.  Serving the file to ui.perfetto.dev (what the reference's stop_recording always does, blocking until the
browser has fetched it) is opt-in here: `profile('traces.json', serve=True)`."""
import socketserver
from contextlib import contextmanager
from http.server import SimpleHTTPRequestHandler

from . import _bindings as B
from .context import _get_ctx


def start_recording():
    B.dsc_traces_record(_get_ctx(), True)


class _PerfettoServer(SimpleHTTPRequestHandler):
    def log_message(self, format, *args):     # noqa: A002
        pass

    def end_headers(self):
        self.send_header('Access-Control-Allow-Origin', '*')
        return super().end_headers()

    def do_GET(self):
        self.server.last_request = self.path
        return super().do_GET()

    def do_POST(self):
        self.send_error(404, 'File not found')


def _serve_traces(traces_file: str, port: int = 9001):
    socketserver.TCPServer.allow_reuse_address = True
    with socketserver.TCPServer(('127.0.0.1', port), _PerfettoServer) as httpd:
        print(f'Open URL in browser: https://ui.perfetto.dev/#!/?url=http://127.0.0.1:{port}/{traces_file}')
        while httpd.__dict__.get('last_request') != '/' + traces_file:
            httpd.handle_request()


def stop_recording(traces_file: str, clear: bool = True, serve: bool = False):
    B.dsc_traces_record(_get_ctx(), False)
    B.dsc_dump_traces(_get_ctx(), traces_file.encode())
    if serve:
        _serve_traces(traces_file)
    if clear:
        B.dsc_clear_traces(_get_ctx())


@contextmanager
def profile(dump_file: str = 'traces.json', serve: bool = False):
    start_recording()
    try:
        yield
    finally:
        stop_recording(dump_file, True, serve)
