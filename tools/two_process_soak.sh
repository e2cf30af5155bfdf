for rep in 1 2 3; do
(timeout 300 python tools/chol_soak.py 3000 4 8 2>&1 | grep -v amdgpu | sed 's/^/A: /' &) ; timeout 300 python tools/chol_soak.py 3000 4 8 2>&1 | grep -v amdgpu | sed 's/^/B: /'; wait; sleep 2
done
