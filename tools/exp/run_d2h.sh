# which engine moves a pinned D2H copy?  (tools/exp/d2h_engine.hip)
cd tools/exp
TL=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
for args in "16 0 0" "16 1 0" "16 0 1" "16 1 1"; do
  echo "== /opt/rocm runtime, args $args"; ./d2h_engine $args 2>&1 | tail -2
  echo "== torch's bundled runtime, args $args"; LD_LIBRARY_PATH=$TL ./d2h_engine $args 2>&1 | tail -2
done
