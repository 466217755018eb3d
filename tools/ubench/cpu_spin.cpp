#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <atomic>
int main(){ for(int T: {1,8,16,32,64,128,256}){ std::atomic<long long> tot{0}; auto t0=std::chrono::steady_clock::now(); std::vector<std::thread> th; for(int t=0;t<T;t++) th.emplace_back([&]{ volatile unsigned long long x=1; long long n=0; auto e=std::chrono::steady_clock::now()+std::chrono::milliseconds(300); while(std::chrono::steady_clock::now()<e){ for(int i=0;i<100000;i++) x=x*6364136223846793005ULL+1; n+=100000;} tot+=n;}); for(auto&x:th)x.join(); double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count(); printf("T=%d: %.2f G iter/s total, %.3f per thread\n",T,tot/dt/1e9,tot/dt/1e9/T);} }
